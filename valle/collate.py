from valle2_amd.collate import ValleARCollate, ValleNARCollate, collate_list, get_collate  # noqa: F401
