"""Model plugins, resolved by `config['model']['name']` as the reference's main.py:12-13,333 does."""
