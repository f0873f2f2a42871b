"""Distiller factory with the reference's contract (distillers/__init__.py:13-41): `create_distiller(opt)` resolves
`<opt.distiller>_distiller.py` and the class whose lower-cased name is `<distiller>distiller`; `get_option_setter`
returns that class's `modify_commandline_options`."""
import importlib


def find_distiller_using_name(distiller_name):
    distiller_filename = 'cat_amd.distillers.' + distiller_name + '_distiller'
    try:
        modellib = importlib.import_module(distiller_filename)
    except ModuleNotFoundError as e:
        raise NotImplementedError('distiller [%s] is not part of the accelerated hot path (inception and spade are)'
                                  % distiller_name) from e
    target = distiller_name.replace('_', '') + 'distiller'
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type):
            return cls
    raise NotImplementedError('In %s.py, there should be a class matching %s in lowercase.' % (distiller_filename, target))


def get_option_setter(distiller_name):
    return find_distiller_using_name(distiller_name).modify_commandline_options


def create_distiller(opt, verbose=True):
    distiller = find_distiller_using_name(opt.distiller)
    instance = distiller(opt)
    if verbose:
        print('distiller [%s] was created' % type(instance).__name__)
    return instance
