"""Import alias so that the reference's import paths (`valle.config`, `valle.models.*`) resolve to
the MI355X implementation in `valle2_amd` — the reference's train_model.py and tests import these."""
