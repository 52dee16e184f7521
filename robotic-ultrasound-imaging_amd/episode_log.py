"""Episode CSV dump in the reference's wire format (`save_data=True`, ultrasound.py:479-509, 552-614, 890-910).

The reference records 24 arrays of `horizon` rows for its single environment and, when the episode ends, writes each one to
`<folder>/<name>_<idx>.csv` (no header, no index, idx = first free integer >= 1) in three folders.  Here the step kernel emits
the same channels for every environment into a [n, 53] device record (`usim_step_io.log_dev`); this class follows one environment,
buffers its rows and writes the same files at the end of its episode.  utils/plot.py / utils/error.py of the reference read them
unchanged."""
import os

import numpy as np

# (folder, file stem, first column, width) in the order of ultrasound.py:584-612
_CHANNELS = [
    ("simulation_data", "ee_pos", 0, 3), ("simulation_data", "ee_goal_pos", 3, 3), ("simulation_data", "ee_vel", 6, 3),
    ("simulation_data", "ee_goal_vel", 9, 1), ("simulation_data", "ee_running_mean_vel", 10, 1), ("simulation_data", "ee_quat", 11, 4),
    ("simulation_data", "ee_goal_quat", 15, 4), ("simulation_data", "ee_diff_quat", 19, 1), ("simulation_data", "ee_z_contact_force", 20, 1),
    ("simulation_data", "ee_z_goal_contact_force", 21, 1), ("simulation_data", "ee_z_running_mean_contact_force", 22, 1),
    ("simulation_data", "ee_z_derivative_contact_force", 23, 1), ("simulation_data", "ee_z_goal_derivative_contact_force", 24, 1),
    ("simulation_data", "is_contact", 25, 1), ("simulation_data", "q_pos", 26, 7), ("simulation_data", "q_torques", 33, 7),
    ("simulation_data", "time", 40, 1),
    ("reward_data", "pos", 41, 1), ("reward_data", "ori", 42, 1), ("reward_data", "vel", 43, 1), ("reward_data", "force", 44, 1),
    ("reward_data", "derivative_force", 45, 1),
    ("policy_data", "action", 46, None),        # width = action_dim
]


def _save(data, folder, stem):
    """ultrasound.py:890-910: first free `<stem>_<idx>.csv`, rows as comma-separated shortest float reprs (pandas default)"""
    os.makedirs(folder, exist_ok=True)
    idx = 1
    path = os.path.join(folder, f"{stem}_{idx}.csv")
    while os.path.exists(path):
        idx += 1
        path = os.path.join(folder, f"{stem}_{idx}.csv")
    with open(path, "w") as f:
        for row in np.atleast_2d(data.T).T if data.ndim == 1 else data:
            f.write(",".join(repr(float(v)) for v in np.atleast_1d(row)) + "\n")
    return path


class EpisodeLogger:
    """Follow environment `env_index` of a UltrasoundVecEnv and dump its episodes like the reference does."""

    def __init__(self, vec_env, env_index=0, root="."):
        self.env, self.i, self.root = vec_env, int(env_index), root
        self.horizon, self.adim = vec_env.horizon, vec_env.action_dim
        if vec_env.step_log is None:
            vec_env.enable_step_log(True)
        self._rows = np.zeros((self.horizon, 53), dtype=np.float64)       # zero-filled like the reference's np.zeros(horizon, ...)
        self.written = []

    def after_step(self, done):
        """Call after every step (before the next one); `done` is that step's done flag of the followed environment."""
        row = self.env.step_log[self.i].cpu().numpy().astype(np.float64)
        t = int(round(row[40] * self.horizon / 100.0))                     # time channel = (timestep - 1) / horizon * 100
        if 0 <= t < self.horizon:
            self._rows[t] = row
        if done:
            self.flush()

    def flush(self):
        for folder, stem, col, width in _CHANNELS:
            w = self.adim if width is None else width
            data = self._rows[:, col:col + w]
            self.written.append(_save(data[:, 0] if w == 1 else data, os.path.join(self.root, folder), stem))
        self._rows[:] = 0.0
