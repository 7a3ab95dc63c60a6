"""Import shim: the package directory is named `zk-cryptography_amd/` (not a valid
Python identifier), so `import zk_cryptography_amd` resolves here and this module
re-exports that directory as a package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "zk-cryptography_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
