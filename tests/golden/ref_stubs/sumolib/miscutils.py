def getFreeSocketPort():
    return 0
