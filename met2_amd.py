"""Import alias: the package directory is named `multicomponent-t2-toolbox_amd` (not a Python
identifier); `import met2_amd` gives the same module object."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("multicomponent-t2-toolbox_amd")
