"""Can two kernel chains share the chip by CU MASK instead of by luck?  (round-4 experiment)

hipExtStreamCreateWithCUMask gives a stream whose kernels only run on the masked CUs.  Probe:
  1. a chip-filling kernel (VGG conv3_2 forward, pconv2) on a full stream, on a half-chip stream (two mask layouts);
  2. chain A = 6 big convs (VGG conv2_2 .. conv4_2 shapes), chain B = 40 small launches (discriminator layer 4-6 shapes +
     BatchNorm passes): A alone, B alone, A then B on one stream, A || B on two unmasked streams, A || B on two
     complementary half-chip streams -- eager and as hipGraphs launched on those streams.
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import csmri_hip
from csmri_hip import ops

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))   # the runtime torch has loaded


def masked_stream(mask_words):
  st = C.c_void_p()
  arr = (C.c_uint32 * len(mask_words))(*mask_words)
  rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(mask_words)), arr)
  assert rc == 0, 'hipExtStreamCreateWithCUMask rc %d' % rc
  return torch.cuda.ExternalStream(st.value)


def conv(cin, cout, k, stride, border, b, h):
  w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device='cuda') * 0.02)
  bias = torch.nn.Parameter(torch.zeros(cout, device='cuda'))
  pad = (1, 1, 1, 1) if k == 3 else ((1, 2, 1, 2) if stride == 1 else (1, 1, 1, 1))
  layer = ops.ConvLayer(w, bias, stride, pad, border, torch.bfloat16)
  x = torch.randn(b, h, h, ops.pad8(cin), device='cuda').to(torch.bfloat16)
  return layer, x


def timeit(fn, reps=20):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e6


def main():
  dev = torch.device('cuda', 0)
  full = [0xffffffff] * 8
  lo = [0xffffffff] * 4 + [0] * 4                 # CUs 0..127
  hi = [0] * 4 + [0xffffffff] * 4                 # CUs 128..255
  even = [0x55555555] * 8                         # every other CU
  odd = [0xaaaaaaaa] * 8
  s_full, s_lo, s_hi, s_even, s_odd = [masked_stream(m) for m in (full, lo, hi, even, odd)]
  plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()

  big = [conv(128, 128, 3, 1, 'zero', 16, 128), conv(128, 256, 3, 1, 'zero', 16, 64), conv(256, 256, 3, 1, 'zero', 16, 64),
         conv(256, 256, 3, 1, 'zero', 16, 64), conv(256, 512, 3, 1, 'zero', 16, 32), conv(512, 512, 3, 1, 'zero', 16, 32)]
  small = [conv(256, 512, 4, 2, 'reflection', 24, 32), conv(512, 1024, 4, 2, 'reflection', 24, 16),
           conv(1024, 1024, 4, 1, 'reflection', 24, 8), conv(128, 256, 4, 2, 'reflection', 24, 64)]

  def chain_a():
    for l, x in big:
      ops.conv_forward(l, x, None, True, 0.0, False, None)

  def chain_b():
    for _ in range(3):
      for l, x in small:
        y, _ = ops.conv_forward(l, x, None, True, 0.2, False, None)
        ops.act_bwd(y, y, 0.2)
        ops.act_bwd(y, y, 0.2)

  def on(stream, fn):
    def run():
      with torch.cuda.stream(stream):
        fn()
    return run

  def both(sa, sb):
    def run():
      ev = torch.cuda.Event(); ev.record()
      sa.wait_event(ev); sb.wait_event(ev)
      with torch.cuda.stream(sa):
        chain_a()
      with torch.cuda.stream(sb):
        chain_b()
      torch.cuda.current_stream().wait_stream(sa)
      torch.cuda.current_stream().wait_stream(sb)
    return run

  l0, x0 = big[2]
  one = lambda: ops.conv_forward(l0, x0, None, True, 0.0, False, None)
  print('one pconv2 launch (VGG conv3_2, B16): default %.1f us | mask full %.1f | mask lo-half %.1f | hi-half %.1f | even CUs %.1f' % (
      timeit(one), timeit(on(s_full, one)), timeit(on(s_lo, one)), timeit(on(s_hi, one)), timeit(on(s_even, one))))
  print('chain A (6 big convs) alone: default %.1f us | lo-half %.1f | even %.1f' % (timeit(chain_a), timeit(on(s_lo, chain_a)), timeit(on(s_even, chain_a))))
  print('chain B (36 small launches) alone: default %.1f us | hi-half %.1f | odd %.1f' % (timeit(chain_b), timeit(on(s_hi, chain_b)), timeit(on(s_odd, chain_b))))
  print('A then B, one stream: %.1f us' % timeit(lambda: (chain_a(), chain_b())))
  print('A || B eager: two plain streams %.1f us | lo/hi halves %.1f | even/odd %.1f' % (
      timeit(both(plain_a, plain_b)), timeit(both(s_lo, s_hi)), timeit(both(s_even, s_odd))))

  # the same as hipGraphs (what the training step replays)
  def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode='thread_local'):
      fn()
    return g
  cap = torch.cuda.Stream()
  ga, gb = capture(chain_a, cap), capture(chain_b, cap)
  gab = capture(lambda: (chain_a(), chain_b()), cap)

  def graphs(sa, sb):
    def run():
      ev = torch.cuda.Event(); ev.record()
      sa.wait_event(ev); sb.wait_event(ev)
      with torch.cuda.stream(sa):
        ga.replay()
      with torch.cuda.stream(sb):
        gb.replay()
      torch.cuda.current_stream().wait_stream(sa)
      torch.cuda.current_stream().wait_stream(sb)
    return run
  print('graphs: A+B in one graph %.1f us | A alone %.1f (lo-half %.1f) | B alone %.1f (hi-half %.1f)' % (
      timeit(gab.replay), timeit(ga.replay), timeit(on(s_lo, ga.replay)), timeit(gb.replay), timeit(on(s_hi, gb.replay))))
  print('graphs A || B: two plain streams %.1f us | lo/hi halves %.1f | even/odd %.1f' % (
      timeit(graphs(plain_a, plain_b)), timeit(graphs(s_lo, s_hi)), timeit(graphs(s_even, s_odd))))


if __name__ == '__main__':
  main()
