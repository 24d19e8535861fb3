pos = None
