"""Import helpers for the golden generators: the reference (/root/reference) imports TensorFlow, OpenCV, lxml, cssutils, shapely, gensim ... at
module level although many of its functions are plain numpy / Python.  None of those packages exists in the build image; `install_stubs()` puts
placeholder modules into sys.modules so that the reference's modules can be IMPORTED and their numpy-only functions CALLED on duck-typed inputs.
A placeholder returns another placeholder for any attribute and raises when it is called where a value is needed -- a function that really
needs the missing package fails loudly instead of returning nonsense.  Used in the BUILD container only; nothing of this travels to the GPU box
(the generators' outputs, tests/golden/*.json, are data)."""
import importlib.abc
import importlib.machinery
import math
import sys
import types

STUBBED = ("tensorflow", "cv2", "lxml", "cssutils", "shapely", "gensim", "kneed", "rasterio", "textdistance", "jpype", "jpype1", "pythonrc",
           "fiona", "skimage", "imageio", "nltk", "flair", "spacy", "fasttext", "Levenshtein", "editdistance", "tqdm_missing", "absl")


class _Placeholder:
    def __init__(self, name):
        self.__dict__["_name"] = name

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Placeholder(self._name + "." + item)

    def __call__(self, *a, **k):
        return _Placeholder(self._name + "()")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):                      # `class X(stub.Something):` -> a plain class
        return (object,)

    def __repr__(self):
        return f"<placeholder {self._name}>"


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Placeholder(self.__name__ + "." + item)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_stubs():
    import numpy as np
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if not hasattr(np, "math"):
        np.math = math                                       # (numpy 2 dropped the alias the reference uses)
    for alias, ty in (("bool", bool), ("int", int), ("float", float), ("object", object)):   # numpy 1.24 dropped these aliases
        if alias not in np.__dict__:
            setattr(np, alias, ty)
    import collections
    import collections.abc
    for name in collections.abc.__all__:              # (Python 3.10 dropped the collections.* aliases)
        if not hasattr(collections, name):
            setattr(collections, name, getattr(collections.abc, name))
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
