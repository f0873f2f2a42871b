"""Model factory with the reference's contract (models/__init__.py:5-51): `create_model(opt)` resolves
`<opt.model>_model.py` and the class whose lower-cased name is `<model>model`.  Built here: the teacher-training steps that share
the distillation step's op set: `pix2pix` and `cycle_gan`."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    try:
        modellib = importlib.import_module('cat_amd.models.' + model_name + '_model')
    except ModuleNotFoundError as e:
        raise NotImplementedError('model [%s] is not part of the accelerated path (pix2pix and cycle_gan are)' % model_name) from e
    target = model_name.replace('_', '') + 'model'
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            return cls
    raise NotImplementedError('In %s_model.py, there should be a subclass of BaseModel matching %s in lowercase.' % (model_name, target))


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt, verbose=True):
    instance = find_model_using_name(opt.model)(opt)
    if verbose:
        print('model [%s] was created' % type(instance).__name__)
    return instance
