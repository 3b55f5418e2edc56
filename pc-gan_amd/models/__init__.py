"""Model registry -- reference models/__init__.py:5-39 (name -> `<name>_model.py`, class
`<Name>Model`, case-insensitive, must subclass BaseModel)."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    module_name = '%s.%s_model' % (__name__, model_name)
    try:
        modellib = importlib.import_module(module_name)
    except ModuleNotFoundError as e:
        if e.name != module_name:
            raise
        raise NotImplementedError('pcgan_amd: model [%s] is outside the MI355X hot path (available: wsgan_emb, wsgan_cycle)'
                                  % model_name)
    target = model_name.replace('_', '') + 'model'
    model = None
    for name, cls in vars(modellib).items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            model = cls
    if model is None:
        print('In %s.py, there should be a subclass of BaseModel with class name that matches %s in lowercase.'
              % (module_name, target))
        exit(0)
    return model


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(opt.model)()
    instance.initialize(opt)
    print('model [%s] was created' % instance.name())
    return instance
