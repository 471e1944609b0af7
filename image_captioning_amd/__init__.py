"""Import alias: the package directory is named `image-captioning_amd`, which is not a Python
identifier, so `import image_captioning_amd` resolves to it through this stub."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "image-captioning_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
