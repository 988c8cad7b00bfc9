def primitive(f):
    return f


def defvjp(*args, **kwargs):
    return None
