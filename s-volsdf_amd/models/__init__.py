from svs_hip.refpath import extend_package_path

# submodules this repository does not provide are found in the reference checkout (svs_hip/refpath.py)
__path__ = extend_package_path(__name__, __path__)

from models.CasMVSNet import CascadeMVSNet  # noqa: E402,F401  (models/__init__.py:2 of the reference)
