def update(*a, **k):
    pass
