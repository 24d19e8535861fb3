def __getattr__(n): return 0
