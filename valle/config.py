from valle2_amd.config import ConfigValle  # noqa: F401
