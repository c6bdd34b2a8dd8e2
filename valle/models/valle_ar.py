from valle2_amd.valle_ar import ValleAR  # noqa: F401
