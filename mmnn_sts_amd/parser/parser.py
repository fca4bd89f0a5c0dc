"""Model factory of the reference's configuration layer (parser/parser.py:21-198), restricted to the fusion path.

`Parser(config).parseConfig()` reads the same YAML schema (config.yaml); `getModel(args)` builds the native DenseNet121 /
TinyDensenet and wraps it into `MultiModalModel` for `--images --preop|--postop` exactly like parser/parser.py:105-182.
Fixes: `--preop` alone yields the standalone clinical MLP; the predictor list may be given as an
integer count (`ClinicalModel.NUM_PREDICTORS`) for synthetic data.  Dataset construction (parser.py:43-97) is host I/O outside
the path; main.py substitutes synthetic patients when no data location is configured.
"""
import yaml

from ..exceptions.exceptions import ConfigurationError, InitializationError
from ..models.densenet import DenseNet121, TinyDensenet
from ..models.mlp import MLP
from ..models.multimodal import MultiModalModel
from ..models.resnet import r3d_18

DEFAULT_CONFIG = {
    "ImageModel": {"name": "densenet121", "modality": "t1t2", "feature_layers": 12, "num_classes": 2, "spatial_dims": 3,
                   "in_channels": 2, "dropout_prob": 0.2},
    "ClinicalModel": {"PRE_OP_PREDICTORS": [f"predictor{i}" for i in range(32)], "POST_OP_PREDICTORS": []},
    "Hyperparameters": {"epochs": 100, "learning_rate": 5e-4, "momentum": 0.9, "weight_decay": 1e-4, "train_batch_size": 2,
                        "test_batch_size": 1, "seed": 42, "num_gpus": 1},
}


class Parser:
    def __init__(self, config_path=None):
        self.config_path = config_path
        self.config = None

    def parseConfig(self):
        if self.config_path is None:
            self.config = {k: dict(v) for k, v in DEFAULT_CONFIG.items()}
        else:
            with open(self.config_path) as f:
                self.config = yaml.safe_load(f)
        im = self.config['ImageModel']
        if im['modality'].lower().startswith('t1t2') and im['in_channels'] != 2:
            raise ConfigurationError('T1T2 ImageModel modality requires 2 input channels - current number of in_channels: {}'.format(im['in_channels']))
        return self.config

    def predictors(self, args):
        cm = self.config['ClinicalModel']
        if 'NUM_PREDICTORS' in cm:
            return [f"predictor{i}" for i in range(int(cm['NUM_PREDICTORS']))]
        p = list(cm['PRE_OP_PREDICTORS'])
        if getattr(args, 'postop', False):
            p += list(cm.get('POST_OP_PREDICTORS', []))
        return p

    def getModel(self, args):
        if self.config is None:
            raise InitializationError('Attempted to load model prior to parsing config parameters, config must be parsed prior to loading model')
        im = self.config['ImageModel']
        name = im['name'].lower()
        clinical = getattr(args, 'preop', False) or getattr(args, 'postop', False)
        if not args.images and clinical:
            return MLP(len(self.predictors(args)), im['num_classes'], im['feature_layers'])
        kw = dict(spatial_dims=im['spatial_dims'], in_channels=im['in_channels'], out_channels=im['num_classes'],
                  feature_channels=im['feature_layers'], dropout_prob=im['dropout_prob'])
        if name.startswith('densenet121'):
            model = DenseNet121(**kw)
        elif name.startswith('tinydensenet'):
            model = TinyDensenet(**kw)
        elif name.startswith('r3d_18'):
            model = r3d_18(im['num_classes'])                       # parser/parser.py:151-152
        else:
            raise ConfigurationError('Model name not recognized: {}\n\tThe MI355X path provides densenet121, tinydensenet and r3d_18'.format(name))
        if args.images and clinical:
            # parser/parser.py:159-160,171-172: only encoders with .backbone / .features can feed the fusion model
            assert name.startswith('tinydensenet') or name.startswith('densenet121'), \
                "Image models used to build multimodal models must be one of 'tinydensenet' or 'densenet121'"
            model = MultiModalModel(model, self.predictors(args), im['num_classes'], im['feature_layers'], blend=getattr(args, 'blend', False))
        return model
