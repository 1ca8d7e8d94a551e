"""Drop-in for the config / optimizer part of embedding_net/utils.py (reference :143-197).

`parse_params` keeps the reference's YAML schema and the layout of the dict it returns (lower-case
section keys; MODEL.input_shape mirrored into GENERATOR and SOFTMAX_PRETRAINING; the optimizer entry of a
section replaced by an optimizer object; `augmentations` keys set).  One structural difference: Keras
optimizers are constructed without parameters, ours need them, so the optimizer object is an
`OptimizerSpec` whose `.build(parameters)` returns the optimizer (embeddingnet_amd/optimizers.py: the
Keras update rules and defaults, one HIP launch per step).
"""
import yaml

# YAML section -> key of the returned dict (reference utils.py:169-195); the last one is optional
SECTIONS = (("DATALOADER", "dataloader"), ("GENERATOR", "generator"), ("MODEL", "model"), ("TRAIN", "train"),
            ("GENERAL", "general"), ("ENCODINGS", "encodings"))
OPTIONAL_SECTION = ("SOFTMAX_PRETRAINING", "softmax")


class OptimizerSpec:
    """What `get_optimizer(name, lr)` returns: the optimizer's rule and learning rate, bound to parameters later."""

    def __init__(self, name, learning_rate):
        self.name, self.learning_rate = name, float(learning_rate)

    @property
    def rule(self):
        """reference utils.py:144-152: 'adam', 'rms_prop', 'radam', anything else is plain SGD."""
        return self.name if self.name in ("adam", "rms_prop", "radam") else "sgd"

    def build(self, parameters):
        from .optimizers import KerasOptimizer
        return KerasOptimizer(parameters, self.rule, self.learning_rate)

    def __repr__(self):
        return f"OptimizerSpec({self.name!r}, lr={self.learning_rate})"


def get_optimizer(name, learning_rate):
    return OptimizerSpec(name, learning_rate)


def _finish_section(section, input_shape):
    """A section that drives training (TRAIN-like or SOFTMAX_PRETRAINING): optimizer name -> object."""
    section['optimizer'] = get_optimizer(section['optimizer'], section['learning_rate'])
    if input_shape is not None:
        section['input_shape'] = input_shape
        # The reference builds albumentations pipelines only under the misspelt key 'augmentations_type'
        # (utils.py:160-164), i.e. never with the shipped configs; image augmentation is outside the hot path.
        section['augmentations'] = None
    return section


def parse_params(filename='configs/road_signs.yml'):
    with open(filename, 'r') as ymlfile:
        cfg = yaml.safe_load(ymlfile)
    params = {key: (cfg[section] if section != "ENCODINGS" else cfg.get(section, {})) for section, key in SECTIONS}
    shape = params['model']['input_shape']
    params['generator']['input_shape'] = shape
    params['generator']['augmentations'] = None
    _finish_section(params['train'], None)
    section, key = OPTIONAL_SECTION
    if section in cfg:
        params[key] = _finish_section(cfg[section], shape)
    return params
