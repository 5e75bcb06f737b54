from .config import Config, get_cfg, init_cfg, print_cfg, save_cfg, update_cfg  # noqa: F401
