"""Import alias for the package directory ``efficient-gnns_amd/`` (the hyphen is the repo-layout contract).

``import efficient_gnns_amd`` resolves to this module, which declares the hyphenated directory as its
package path and then runs the package's ``__init__``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "efficient-gnns_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
