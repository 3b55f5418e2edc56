"""Name -> class lookup shared by the model and the dataset plugin registries.

The reference resolves `--model X` / `--dataset_mode X` by importing `<pkg>/X_<kind>.py` and picking the subclass whose
lower-cased name is `X` without underscores + `<kind>` (models/__init__.py:5-27, data/__init__.py:5-30); an unknown
module is an ImportError there.  Same rule here, one implementation, and a NotImplementedError that names what IS
available when the plugin is outside the MI355X hot path."""
import importlib
import pkgutil


def available(package, kind):
    pkg = importlib.import_module(package)
    suffix = '_' + kind
    return sorted(m.name[:-len(suffix)] for m in pkgutil.iter_modules(pkg.__path__)
                  if m.name.endswith(suffix) and not m.name.startswith('base'))


def find_plugin(package, kind, name, base_cls):
    """class `<Name><Kind>` (case-insensitive, underscores dropped) from module `<package>.<name>_<kind>`."""
    module_name = '%s.%s_%s' % (package, name, kind)
    try:
        module = importlib.import_module(module_name)
    except ModuleNotFoundError as e:
        if e.name != module_name:
            raise
        raise NotImplementedError('pcgan_amd: %s [%s] is outside the MI355X hot path (available: %s)'
                                  % (kind, name, ', '.join(available(package, kind))))
    wanted = (name.replace('_', '') + kind).lower()
    for attr, cls in vars(module).items():
        if attr.lower() == wanted and isinstance(cls, type) and issubclass(cls, base_cls):
            return cls
    # the reference prints this and exits with status 0 (models/__init__.py:22-24)
    print('In %s.py, there should be a subclass of %s with class name that matches %s in lowercase.'
          % (module_name, base_cls.__name__, wanted))
    raise SystemExit(0)
