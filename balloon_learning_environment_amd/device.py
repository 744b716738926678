"""Thin torch plumbing: device tensors, current stream, pointers.  torch is used for
device memory, streams and torch.distributed only -- all arithmetic is in libble_hip.so."""
import torch

from balloon_learning_environment_amd import _abi

_TORCH_DTYPES = {'float32': torch.float32, 'int64': torch.int64, 'int32': torch.int32, 'uint8': torch.uint8}


def torch_dtype(np_dtype):
  import numpy as np
  return _TORCH_DTYPES[np.dtype(np_dtype).name]


def require_gpu(device) -> torch.device:
  device = torch.device(device)
  if device.type != 'cuda' or not torch.cuda.is_available():
    raise RuntimeError('balloon_learning_environment_amd needs a HIP device (torch device "cuda[:i]"); '
                       'there is no CPU path')
  if device.index is None:
    device = torch.device('cuda', torch.cuda.current_device())
  return device


def on_own_device(method):
  """Decorator for methods of objects with a `.device`: launches go to the stream of the object's OWN device, so that
  device is made current for the call (a kernel launched while another device is current would be enqueued with the
  wrong context).  No-op -- one integer compare -- when it already is."""
  import functools

  @functools.wraps(method)
  def wrapped(self, *args, **kwargs):
    if torch.cuda.current_device() == self.device.index:
      return method(self, *args, **kwargs)
    with torch.cuda.device(self.device):
      return method(self, *args, **kwargs)
  return wrapped


def stream_ptr(device) -> int:
  return torch.cuda.current_stream(device).cuda_stream


def ptr(t) -> int:
  return 0 if t is None else t.data_ptr()


def state_struct(tensors, episode_cache=None):
  """BleStateF32 from {field: torch tensor} (contiguous, right dtype, same device); `episode_cache`: the optional
  [EPISODE_CACHE_ROWS, n] float64 tensor of per-episode derived constants (zero-initialised; opaque)."""
  ptrs = {}
  n = None
  for name in _abi.FIELD_NAMES:
    t = tensors[name]
    assert t.is_contiguous() and t.dtype == torch_dtype(_abi.FIELD_DTYPES[name]), name
    ptrs[name] = t.data_ptr()
    n = t.numel()
  if episode_cache is not None:
    assert episode_cache.is_contiguous() and episode_cache.dtype == torch.float64
    assert tuple(episode_cache.shape) == (_abi.EPISODE_CACHE_ROWS, n)
  return _abi.state_struct(ptrs, 0 if episode_cache is None else episode_cache.data_ptr())
