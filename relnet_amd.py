"""Import alias: the package directory is named after the reference repository
(`relation-networks-for-object-detection_amd`), which is not a valid Python identifier;
`import relnet_amd` resolves to it."""
import importlib
import sys

_pkg = importlib.import_module('relation-networks-for-object-detection_amd')
sys.modules[__name__] = _pkg
