"""Drop-in for the config / optimizer part of embedding_net/utils.py (reference :143-197).

parse_params keeps the reference's YAML schema and returned dict layout.  The one structural
difference: Keras optimizers are constructed without parameters, torch optimizers need them, so
params['train']['optimizer'] is an OptimizerSpec whose .build(parameters) returns the torch optimizer
with the Keras defaults (Adam/RMSprop/RAdam epsilon 1e-7, RMSprop rho .9, plain SGD).
"""
import yaml


class OptimizerSpec:
    def __init__(self, name, learning_rate):
        self.name, self.learning_rate = name, float(learning_rate)

    def build(self, parameters):
        import torch
        lr = self.learning_rate
        if self.name == 'adam':
            return torch.optim.Adam(parameters, lr=lr, betas=(0.9, 0.999), eps=1e-7)
        if self.name == 'rms_prop':
            return torch.optim.RMSprop(parameters, lr=lr, alpha=0.9, eps=1e-7)
        if self.name == 'radam':
            return torch.optim.RAdam(parameters, lr=lr, betas=(0.9, 0.999), eps=1e-7)
        return torch.optim.SGD(parameters, lr=lr)

    def __repr__(self):
        return f"OptimizerSpec({self.name!r}, lr={self.learning_rate})"


def get_optimizer(name, learning_rate):
    return OptimizerSpec(name, learning_rate)


def parse_params(filename='configs/road_signs.yml'):
    with open(filename, 'r') as ymlfile:
        cfg = yaml.safe_load(ymlfile)

    # The reference only builds augmentations when the (misspelt) key 'augmentations_type' exists
    # (utils.py:160-164), i.e. never with the shipped configs; image augmentation is outside the hot path.
    augmentations = None

    optimizer = get_optimizer(cfg['TRAIN']['optimizer'],
                              cfg['TRAIN']['learning_rate'])

    params_dataloader = cfg['DATALOADER']
    params_generator = cfg['GENERATOR']
    params_model = cfg['MODEL']
    params_train = cfg['TRAIN']
    params_general = cfg['GENERAL']
    params_encodings = cfg.get('ENCODINGS', {})

    params_generator['input_shape'] = params_model['input_shape']
    params_train['optimizer'] = optimizer
    params_generator['augmentations'] = augmentations

    params = {'dataloader': params_dataloader,
              'generator': params_generator,
              'model': params_model,
              'train': params_train,
              'general': params_general,
              'encodings': params_encodings}

    if 'SOFTMAX_PRETRAINING' in cfg:
        params_softmax = cfg['SOFTMAX_PRETRAINING']
        params_softmax['augmentations'] = augmentations
        params_softmax['input_shape'] = params_model['input_shape']
        params_softmax['optimizer'] = get_optimizer(cfg['SOFTMAX_PRETRAINING']['optimizer'],
                                                    cfg['SOFTMAX_PRETRAINING']['learning_rate'])
        params['softmax'] = params_softmax

    return params
