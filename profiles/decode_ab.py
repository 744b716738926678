"""A/B of ble_decode_flow_fields_f32 variants inside one gpurun call: build_ab/libble_dec<G>.so built with
-DBLE_DECODE_GROUPS=<G> (profiles/build_variant.sh dec<G> -DBLE_DECODE_GROUPS=<G>); 32 768 grids per launch (10.4 GB written).
  python profiles/decode_ab.py 4 5 6 7 8 11"""
import ctypes, os, sys, statistics
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 32768
flow = torch.randn(n, 7 * 7 * 90, device='cuda')
grid = torch.empty(n, 21 * 21 * 90 * 2, device='cuda')
ref = None
for g in sys.argv[1:]:
  lib = ctypes.CDLL(os.path.join(root, 'build_ab', f'libble_dec{g}.so'))
  lib.ble_decode_flow_fields_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
  ts = []
  for rep in range(7):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.ble_decode_flow_fields_f32(flow.data_ptr(), grid.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
  if ref is None:
    ref = grid.clone()
  same = bool(torch.equal(ref, grid))
  ms = statistics.median(ts[1:])
  print(f'groups {g}: {ms:.3f} ms per 32 768 grids = {n * 317520 / ms / 1e9:.2f} TB/s written; identical to the first variant: {same}')
