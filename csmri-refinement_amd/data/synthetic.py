"""Synthetic undersampled k-space batches following the reference's forward model.

  img (real, [0,1])  --fft2 ortho-->  k_full  --mask-->  kspace  --ifft2 ortho-->  inp
  (rec_transforms.py:18-57, compressed_sensing.py:460-512, dnn_io.py:4-61,
   myImageTransformations.py:1196-1238); Cartesian variable-density mask with 8 centre
  lines, constant along W, ifftshift-ed (compressed_sensing.py:82-123).
All arithmetic in numpy complex128 like the reference, then packed to float32."""
import numpy as np
import torch


def cartesian_mask(shape, acc, sample_n=8, rng=None):
  """(N, nx, ny) 0/1 mask, un-centred.  Row pdf: gaussian + uniform floor, the
  sample_n centre rows forced, nx//acc rows in total."""
  rng = np.random if rng is None else rng
  n, nx, ny = shape
  grid = np.arange(nx) - nx / 2
  pdf = np.exp(-(0.5 / (nx / 10.) ** 2) * grid ** 2) + (nx / (2. * acc)) / nx
  n_lines = nx // acc
  lo, hi = nx // 2 - sample_n // 2, nx // 2 + sample_n // 2
  if sample_n:
    pdf[lo:hi] = 0
    pdf /= pdf.sum()
    n_lines -= sample_n
  rows = np.zeros((n, nx))
  for i in range(n):
    rows[i, rng.choice(nx, int(n_lines), False, pdf)] = 1
  if sample_n:
    rows[:, lo:hi] = 1
  mask = np.repeat(rows[:, :, None], ny, axis=2)
  return np.fft.ifftshift(mask, axes=(-1, -2))


def radial_mask(shape, n_lines, rng=None, golden_angle=True, rand=True, centred=False):
  """Radial undersampling (BASELINE config 5; reference compressed_sensing.py:568-647 as the training
  transform calls it, myImageTransformations.py:63-70: ``acceleration_factor`` is the NUMBER OF SPOKES).
  Every spoke is a line through the k-space centre gridded to the nearest Cartesian sample; golden-angle
  spokes continue across the slices of a batch from one random start angle, uniform spokes get a random
  offset per slice.  Returns an integer (N, nx, nx) 0/1 mask, ifftshift-ed unless ``centred``."""
  rng = np.random if rng is None else rng
  n, nx, ny = shape
  assert nx == ny, 'square slices only'
  start = np.pi * rng.random() if rand else 0.0
  if golden_angle:
    step = np.pi / ((1.0 + np.sqrt(5.0)) / 2.0)
    angles = start + step * np.arange(n_lines * n)
  else:
    angles = np.tile(np.arange(0, np.pi, np.pi / n_lines), n)
    angles = angles + np.repeat(rng.random(n) * np.pi / n_lines, n_lines)
  radius = np.arange(-nx / 2, nx / 2, 1.0)
  # 1-based nearest sample of every (radius, spoke), wrapped into [1, nx]
  col = np.round(np.outer(radius, np.cos(angles)) + 0.5) + nx / 2
  row = np.round(np.outer(radius, np.sin(angles)) + 0.5) + ny / 2
  col = np.where(col > nx, col - nx, col); col = np.where(col < 1, col + nx, col)
  row = np.where(row > ny, row - ny, row); row = np.where(row < 1, row + ny, row)
  mask = np.zeros((n, nx, ny), dtype=int)
  slice_of = np.repeat(np.arange(n), n_lines * nx)
  mask[slice_of, col.T.reshape(-1).astype(int) - 1, row.T.reshape(-1).astype(int) - 1] = 1
  return mask if centred else np.fft.ifftshift(mask, axes=(-2, -1))


def synth_batch_radial(b, h, w, spokes=70, seed=0):
  """A batch as synth_batch() with golden-angle radial masks (one random start angle per batch)."""
  m = radial_mask((b, h, w), spokes, rng=np.random.RandomState(seed + 7919))
  parts = []
  for i in range(b):
    img = phantom(h, w, seed + 1000 + i)
    k_u = m[i] * np.fft.fft2(img.astype(np.complex128), norm='ortho')
    x_u = np.fft.ifft2(k_u, norm='ortho')
    parts.append((_pack(x_u), _pack(k_u), _pack(m[i] * (1 + 1j)), _pack(img.astype(np.complex128))))
  return {k: torch.from_numpy(np.stack([p[j] for p in parts]))
          for j, k in enumerate(('inp', 'kspace', 'mask', 'target'))}


def phantom(h, w, seed):
  """Seeded smooth random field plus a few ellipses, strictly positive, max 1."""
  rs = np.random.RandomState(seed)
  spec = np.fft.fft2(rs.rand(h, w))
  fy, fx = np.fft.fftfreq(h)[:, None], np.fft.fftfreq(w)[None, :]
  img = np.real(np.fft.ifft2(spec * np.exp(-(fy ** 2 + fx ** 2) * (h * 0.35) ** 2)))
  img = (img - img.min()) / (img.max() - img.min() + 1e-12)
  yy, xx = np.mgrid[0:h, 0:w]
  for _ in range(4):
    cy, cx = rs.uniform(0.25, 0.75) * h, rs.uniform(0.25, 0.75) * w
    ry, rx = rs.uniform(0.05, 0.25) * h, rs.uniform(0.05, 0.25) * w
    img = img + rs.uniform(0.2, 0.8) * (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1)
  img = img + 0.02
  return img / np.max(np.abs(img))


def _pack(z):
  return np.stack((np.real(z), np.imag(z))).astype(np.float32)


def synth_sample(h, w, acc, seed, sample_n=8):
  img = phantom(h, w, seed)
  mask = cartesian_mask((1, h, w), acc, sample_n, np.random.RandomState(seed + 7919))[0]
  k_u = mask * np.fft.fft2(img.astype(np.complex128), norm='ortho')
  x_u = np.fft.ifft2(k_u, norm='ortho')
  return _pack(x_u), _pack(k_u), _pack(mask * (1 + 1j)), _pack(img.astype(np.complex128))


def synth_batch(b, h, w, acc=4, seed=0, sample_n=8, first=0):
  """Samples first .. first + b - 1 of the batch seeded ``seed`` (a sample depends on its index only: rank r of a
  data-parallel job synthesises rows r * b .. of the global batch and nothing else)."""
  parts = [synth_sample(h, w, acc, seed + 1000 + first + i, sample_n) for i in range(b)]
  return {k: torch.from_numpy(np.stack([p[j] for p in parts]))
          for j, k in enumerate(('inp', 'kspace', 'mask', 'target'))}


class SyntheticLoader(object):
  """Minimal DataLoader stand-in: ``len``, ``iter``, ``batch_size``.  Batches are
  generated once (``distinct`` of them, pinned) and cycled.  ``shard = (rank, world)``: ``batch_size`` is the GLOBAL
  batch and this loader holds only rank's contiguous share of it (``presharded``: the runner's _request_data does not
  cut it again) -- the same rows dist_utils.shard_batch would cut out of the global batch, without every rank
  synthesising all of it."""

  def __init__(self, batch_size, h, w, num_batches, acc=4, seed=0, distinct=2, pin=True, shard=None):
    self.batch_size, self.num_batches = batch_size, num_batches
    self.presharded = shard is not None and shard[1] > 1
    first, rows = 0, batch_size
    if self.presharded:
      r, n = shard
      assert batch_size % n == 0, 'global batch {} not divisible by {} ranks'.format(batch_size, n)
      rows = batch_size // n
      first = r * rows
    self.batches = []
    for i in range(max(1, min(distinct, num_batches))):
      bt = synth_batch(rows, h, w, acc, seed + 100000 * i, first=first)
      if pin and torch.cuda.is_available():
        bt = {k: v.pin_memory() for k, v in bt.items()}
      self.batches.append(bt)

  def __len__(self):
    return self.num_batches

  def __iter__(self):
    for i in range(self.num_batches):
      yield self.batches[i % len(self.batches)]


def synth_batch_device(b, h, w, acc=4, seed=0, sample_n=8, device=None):
  """Same samples as synth_batch with the forward model on the GPU: the phantom
  and the mask rows are drawn on the host exactly as above (tiny, RNG-defined), the two FFTs per
  slice run in ``csmri_undersample`` in fp32.  Returns NHWC device tensors: inp, kspace, target
  interleaved complex [B,H,W,2] fp32 and mask in the reference's batch layout [B,2,H,W]."""
  from csmri_hip import ops
  device = device or torch.device('cuda', torch.cuda.current_device())
  imgs, masks = [], []
  for i in range(b):
    s = seed + 1000 + i
    imgs.append(phantom(h, w, s))
    masks.append(cartesian_mask((1, h, w), acc, sample_n, np.random.RandomState(s + 7919))[0])
  img = torch.from_numpy(np.stack(imgs).astype(np.float32)).to(device)
  mask = torch.from_numpy(np.stack(masks).astype(np.float32)).to(device)
  target = torch.stack((img, torch.zeros_like(img)), dim=-1).contiguous()       # [B,H,W,2]
  kspace, inp = ops.undersample(target, (mask != 0).to(torch.uint8))
  return {'inp': inp, 'kspace': kspace, 'target': target,
          'mask': torch.stack((mask, mask), dim=1).contiguous()}
