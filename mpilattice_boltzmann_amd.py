"""Import alias: the package directory is `mpilattice-boltzmann_amd/` (hyphen, as the project is
named), which the `import` statement cannot spell.  `import mpilattice_boltzmann_amd` gives that
package."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("mpilattice-boltzmann_amd")
