class Resample:
    def __init__(self, *a, **k):
        raise NotImplementedError
