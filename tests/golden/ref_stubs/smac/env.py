class MultiAgentEnv:
    pass
