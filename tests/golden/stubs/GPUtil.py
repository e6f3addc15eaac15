def getGPUs():
    return []
