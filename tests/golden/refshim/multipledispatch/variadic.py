def isvariadic(obj):
    return False
