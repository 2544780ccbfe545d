"""Data side of the hot path.  The reference trains on a proprietary dataset
(README.md:7-9); this package only provides the synthetic generator that emits the
same batch dict (keys inp/kspace/mask/target, each [B,2,H,W] fp32,
scar_segmentation.py:212-218) from the reference's own numpy forward model."""
from data.synthetic import SyntheticLoader, synth_batch, cartesian_mask  # noqa: F401


def load_dataset(conf, data_dir, name, fold):
  if name not in ('synthetic', 'ScarSeg'):
    raise ValueError('Unknown dataset {}'.format(name))
  raise NotImplementedError('dataset loading goes through data.synthetic.SyntheticLoader; the '
                            'ScarSeg data is proprietary and not part of this build')
