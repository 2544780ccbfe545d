"""Weight initialisation by scheme name -- same scheme vocabulary and override
precedence as the reference (models/weight_inits.py:5-114): defaults <
model.weight_init_params() < user ``weight_init`` config; per-module overrides
keyed by the module object win over class-wide keys."""
import torch.nn.init as init

DEFAULT_INITS = {
    'conv_weight': ('he_normal', 0.0),
    'conv_bias': ('constant', 0.0),
    'batchnorm_weight': ('constant', 1.0),
    'batchnorm_bias': ('constant', 0.0),
}


def _apply_scheme(scheme, tensor):
  name = scheme[0] if isinstance(scheme, (tuple, list)) else scheme
  if name == 'torch_default':
    return
  if name == 'zero':
    init.constant_(tensor, 0.0)
  elif name == 'constant':
    init.constant_(tensor, scheme[1])
  elif name == 'normal':
    init.normal_(tensor, mean=scheme[1], std=scheme[2])
  elif name == 'uniform':
    init.uniform_(tensor, a=scheme[1], b=scheme[2])
  elif name.startswith('xavier'):
    gain = scheme[1]
    if isinstance(gain, str):
      gain = init.calculate_gain(gain)
    # NB: only the exact name 'xavier_normal' is normal; plain 'xavier' is uniform
    if name == 'xavier_normal':
      init.xavier_normal_(tensor, gain=gain)
    else:
      init.xavier_uniform_(tensor, gain=gain)
  elif name.startswith('he'):
    a = scheme[1] if isinstance(scheme, (tuple, list)) else 0.0
    if name == 'he_normal':
      init.kaiming_normal_(tensor, a=a)
    else:
      init.kaiming_uniform_(tensor, a=a)
  elif name == 'orthogonal':
    gain = scheme[1] if isinstance(scheme, (tuple, list)) else 1.0
    if isinstance(gain, str):
      gain = init.calculate_gain(gain, scheme[2] if len(scheme) > 2 else None)
    init.orthogonal_(tensor, gain=gain)
  else:
    raise ValueError('Unknown weight init {}'.format(name))


def initialize_weights(model, user_weight_init=None):
  user_weight_init = user_weight_init or {}
  table = dict(DEFAULT_INITS)
  table.update(model.weight_init_params(user_weight_init))
  table.update(user_weight_init)
  for m in model.modules():
    kind = getattr(m, 'kind', None)
    if kind is None:
      continue
    if m in table:                       # per-module override: only listed fields
      w_s, b_s = table[m].get('weight'), table[m].get('bias')
    else:
      w_s, b_s = table.get(kind + '_weight'), table.get(kind + '_bias')
    if w_s is not None and m.weight is not None:
      _apply_scheme(w_s, m.weight.data)
    if b_s is not None and getattr(m, 'bias', None) is not None:
      _apply_scheme(b_s, m.bias.data)
