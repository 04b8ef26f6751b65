"""Import alias for the ``citlab-article-separation-new_amd/`` package directory.

The package directory carries the reference repository's name (with hyphens), which
is not a valid Python identifier.  This alias package extends its ``__path__`` with
that directory so that ``import citlab_article_separation_new_amd.<module>`` works.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "citlab-article-separation-new_amd")
__path__.append(_real)

PACKAGE_DIR = _real
CSRC_DIR = _os.path.join(_real, "csrc")
