"""The import switch of INTEGRATION.md section 1: make the reference's own import statements resolve to this package.

The reference's training script binds the hot path by module name (scripts/train.py:15-20):

    from data.args import get_args, print_args, init_args
    from models.meta_models import AG2VideoModel, MetaDiscriminatorModel
    from models.spade_models.loss_model import LossModel
    from models.utils import batch_to
    from models.spade_models.networks.sync_batchnorm import DataParallelWithCallback

``install()`` registers every module of ``ag2video_amd.models`` (and ``ag2video_amd.data.args``) in ``sys.modules`` under the
reference's names -- the SAME module objects, not second copies, so ``isinstance`` checks and module-level state (timing scopes,
the synchronised-BN switch) are shared between the two names.  ``uninstall()`` removes exactly what ``install()`` added.
Names the hot path does not own (``data.dataset_params``, ``models.vis``, ``models.metrics``, ``evaluation.*``) are left to the
reference tree.
"""
import importlib
import pkgutil
import sys

_INSTALLED = {}


def _walk(pkg_name):
    pkg = importlib.import_module(pkg_name)
    yield pkg_name, pkg
    if hasattr(pkg, '__path__'):
        for info in pkgutil.walk_packages(pkg.__path__, pkg_name + '.'):
            yield info.name, importlib.import_module(info.name)


def install(force=False):
    """Alias ``models[.*]`` and ``data.args`` to this package's modules.  Refuses to shadow an already imported module of another
    origin unless ``force``.  Returns the list of aliased names."""
    plan = {}
    for name, mod in _walk('ag2video_amd.models'):
        plan['models' + name[len('ag2video_amd.models'):]] = mod
    args_mod = importlib.import_module('ag2video_amd.data.args')
    plan['data.args'] = args_mod
    for alias, mod in plan.items():
        cur = sys.modules.get(alias)
        if cur is not None and cur is not mod and not force:
            raise RuntimeError("ag2video_amd.dropin.install: %r is already imported from %s" % (alias, getattr(cur, '__file__', '?')))
    for alias, mod in plan.items():
        if sys.modules.get(alias) is not mod:
            _INSTALLED[alias] = sys.modules.get(alias)
            sys.modules[alias] = mod
    # `from data.args import ...` needs a parent package named `data`: the reference's own one when its tree is on sys.path (then
    # only the attribute is replaced), otherwise this package's
    parent = sys.modules.get('data')
    if parent is None:
        parent = importlib.import_module('ag2video_amd.data')
        _INSTALLED.setdefault('data', None)
        sys.modules['data'] = parent
    if getattr(parent, 'args', None) is not args_mod:
        _INSTALLED.setdefault('data:args', getattr(parent, 'args', None))
        parent.args = args_mod
    return sorted(plan)


def uninstall():
    for alias, prev in list(_INSTALLED.items()):
        if alias == 'data:args':
            parent = sys.modules.get('data')
            if parent is not None:
                if prev is None:
                    try:
                        delattr(parent, 'args')
                    except AttributeError:
                        pass
                else:
                    parent.args = prev
        elif prev is None:
            sys.modules.pop(alias, None)
        else:
            sys.modules[alias] = prev
    _INSTALLED.clear()
