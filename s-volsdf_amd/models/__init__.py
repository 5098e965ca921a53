from models.CasMVSNet import CascadeMVSNet  # noqa: F401
