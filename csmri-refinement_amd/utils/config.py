"""Configuration: attribute bag over JSON, same key semantics as the reference's
utils/config.py (``#include`` / ``include`` composition, ``seed`` -> private
field, get_attr/has_attr/to_param_dict/from_dict/update/serialize) so that the
shipped configs/*.json load unchanged.  Reference: utils/config.py:7-250."""
import json
import os

_TYPE_TAG = '__type__'


class Configuration(object):
  def __init__(self):
    self._seed = 0
    self._src_file = None
    self.__dict__[_TYPE_TAG] = str(type(self))

  # -- construction ----------------------------------------------------------
  @staticmethod
  def from_dict(dictionary, parent_config=None):
    if isinstance(dictionary, Configuration):
      return dictionary
    conf = Configuration()
    conf.__dict__.update(dictionary)
    if parent_config is not None:
      conf._seed = parent_config._seed
      conf._src_file = parent_config._src_file
    return conf

  @staticmethod
  def from_json(src):
    def hook(obj):
      merged = {}
      includes = obj.pop('#include', None)
      if includes is not None:
        for path in (includes if isinstance(includes, list) else [includes]):
          if not os.path.isabs(path):
            path = os.path.join(os.path.dirname(src), path)
          merged.update(Configuration.from_json(path).__dict__)
      if 'seed' in obj:
        merged['_seed'] = obj.pop('seed')
      merged.update(obj)
      if obj.get(_TYPE_TAG) == str(Configuration):
        return Configuration.from_dict(merged)
      return merged

    with open(src, 'r') as f:
      conf = json.load(f, object_hook=hook)
    if isinstance(conf, dict):
      conf = Configuration.from_dict(conf)
    conf._src_file = src
    if hasattr(conf, 'include'):
      for key, path in conf.include.items():
        if not os.path.isabs(path):
          path = os.path.join(os.path.dirname(src), path)
        sub = Configuration.from_json(path)
        if key == '':
          conf.__dict__ = dict(**sub.__dict__, **conf.__dict__)
        else:
          saved = conf.get_attr(key, default=None)
          conf.__dict__[key] = sub.__dict__
          if isinstance(conf.__dict__[key], dict) and isinstance(saved, dict):
            conf.__dict__[key].update(saved)
      del conf.__dict__['include']
    return conf

  # -- access ----------------------------------------------------------------
  @property
  def seed(self):
    return self._seed

  @property
  def file(self):
    return self._src_file

  def has_attr(self, key):
    return hasattr(self, key)

  def get_attr(self, key, default=None, alternative=None):
    if hasattr(self, key):
      return getattr(self, key)
    if alternative is not None:
      value = self.get_attr(alternative)
      if value is None:
        raise ValueError('Configuration did not contain {} or alternative {}'.format(key, alternative))
      return value
    return default

  def to_param_dict(self, required_params=(), optional_params=(), key_renames=None):
    key_renames = key_renames or {}
    params = {}
    for key in required_params:
      value = self.get_attr(key)
      assert value is not None, 'Parameter {} is marked as required'.format(key)
      params[key] = value
    if isinstance(optional_params, dict):
      for key, default in optional_params.items():
        params[key] = self.get_attr(key, default=default)
    else:
      for key in optional_params:
        value = self.get_attr(key)
        if value is not None:
          params[key] = value
    return {key_renames.get(k, k): v for k, v in params.items()}

  def update(self, values_by_keys):
    """--conf k=v overrides with str -> bool/int/float/list coercion
    (reference utils/config.py:108-149)."""
    def convert(s):
      if (s.startswith('[') and s.endswith(']')) or (s.startswith('(') and s.endswith(')')):
        return [convert(e.strip()) for e in s[1:-1].split(',')]
      if s == 'False':
        return False
      if s == 'True':
        return True
      for cast in (int, float):
        try:
          return cast(s)
        except ValueError:
          pass
      return s

    for key, value in values_by_keys.items():
      value = convert(value)
      if key == 'seed':
        self.__dict__['_seed'] = value
      else:
        self.__dict__[key] = value

  def serialize(self, dst):
    with open(dst, 'w') as f:
      json.dump(self.__dict__, f, default=lambda o: o.__dict__, indent=2)

  def __str__(self):
    return 'Configuration object\n' + ''.join('  {}: {}\n'.format(k, v) for k, v in self.__dict__.items())
