"""Alias package: `embedding_net.<module>` IS `embeddingnet_amd.<module>`, so code written against
RocketFlash/EmbeddingNet's import paths (reference tools/train.py:9-15: `from embedding_net.models import TripletNet`,
`from embedding_net.utils import parse_params`, ...) picks up the MI355X hot path.  A module is imported when it is first
asked for (PEP 562 attribute hook for `embedding_net.models`, a finder for `import embedding_net.models`)."""
import importlib
import importlib.abc
import importlib.util
import sys

_MODULES = ("backbones", "datagenerators", "losses_and_accuracies", "models", "parallel", "train_step", "utils")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        pkg, _, name = fullname.partition(".")
        if pkg == __name__ and name in _MODULES:
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("embeddingnet_amd." + spec.name.partition(".")[2])    # the same module object

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):
    if name in _MODULES:
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
