def configure(name, defaults):
    """No-op: the golden generator sets config attributes directly."""
    return None
