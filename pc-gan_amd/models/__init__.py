"""Model plugin registry with the reference's three entry points (models/__init__.py:5-39):
find_model_using_name / get_option_setter / create_model."""
from .base_model import BaseModel
from ..util.registry import find_plugin


def find_model_using_name(model_name):
    return find_plugin(__name__, 'model', model_name, BaseModel)


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    model = find_model_using_name(opt.model)()
    model.initialize(opt)
    print('model [%s] was created' % model.name())
    return model
