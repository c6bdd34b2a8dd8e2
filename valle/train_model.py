from valle2_amd.train_model import train  # noqa: F401
