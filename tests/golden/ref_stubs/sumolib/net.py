def readNet(*a, **k):
    raise RuntimeError("sumolib stub")
