"""PyMARL adapter (reference: pytsc/wrappers/pymarl.py:5-86): `step` returns the
raw (reward, done, info) triple, `reset` returns (obs, state)."""
from pytsc_amd.wrappers.epymarl import EPyMARLTrafficSignalNetwork


class PyMARLTrafficSignalNetwork(EPyMARLTrafficSignalNetwork):
    def get_env_info(self):
        info = super().get_env_info()
        info.pop("obs_info", None)
        info.pop("lane_features", None)
        info.update({"n_domains": 1, "domain_classes": ["baseline"]})
        return info

    def reset(self):
        self.tsc_env.episode_count += 1
        if self.tsc_env.episode_over:
            self.tsc_env.restart()
        return self.get_obs(), self.get_state()

    def step(self, actions):
        return self.tsc_env.step(actions)
