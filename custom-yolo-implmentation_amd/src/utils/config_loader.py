"""config.yaml loader (reference: src/utils/config_loader.py:3-6)."""
import yaml


def load_config(config_path="config.yaml"):
    with open(config_path, "r") as fh:
        return yaml.safe_load(fh)
