class DictConfig(dict):
    """Attribute + item access nested dict (hand-built trees only; no interpolation)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = DictConfig(v) if isinstance(v, dict) and not isinstance(v, DictConfig) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, DictConfig) else v) for k, v in self.items()}
