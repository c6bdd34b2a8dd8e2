from valle2_amd.valle_nar import ValleNAR  # noqa: F401
