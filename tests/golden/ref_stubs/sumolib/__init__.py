def checkBinary(x):
    return x
