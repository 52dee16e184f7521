"""Error metrics of the reference's evaluation scripts (src/utils/error.py:148-191), computed on the device for
EVERY environment of a batch from the per-step episode record (`usim_step_io.log_dev`, [n, 53]).

The reference reads the CSV files of one finished episode (`horizon` rows; rows after an early termination stay zero,
ultrasound.py:479-509) and writes one number per metric to `error_data/<model_name>/<metric>.csv`:

    x_pos_mse, y_pos_mse        error.py:33-52     mean over rows of (ee_pos - ee_goal_pos)^2, x and y
    force_mse, mean_force_mse   error.py:55-72     (Fz - goal)^2, (running mean Fz - goal)^2
    der_force_mse               error.py:75-89
    velocity_mse, mean_velocity_mse   error.py:92-110   (|ee_vel| - goal)^2, (running mean speed - goal)^2
    quat_diff_mean              error.py:145-156   mean of the quaternion distance channel
    pos/ori/force/der_force/vel_reward_mean   error.py:113-142

Every one of them is a sum over the rows of the episode divided by `horizon` (zero rows add nothing), so an accumulator per environment
and metric reproduces them without materialising the rows: `update()` after every step, finished episodes land in `last`."""
import os

import numpy as np
import torch

METRICS = ("x_pos_mse", "y_pos_mse", "force_mse", "mean_force_mse", "der_force_mse", "velocity_mse", "mean_velocity_mse", "quat_diff_mean",
           "pos_reward_mean", "ori_reward_mean", "force_reward_mean", "der_force_reward_mean", "vel_reward_mean")


def step_terms(log):
    """[n, 53] episode record of one step (column layout: episode_log._CHANNELS) -> [n, 13] summands of METRICS"""
    sq = lambda a, b: (a - b) ** 2
    speed = torch.linalg.vector_norm(log[:, 6:9], dim=1)                          # error.py:104 np.linalg.norm per row
    return torch.stack([
        sq(log[:, 0], log[:, 3]), sq(log[:, 1], log[:, 4]),                        # ee_pos vs ee_goal_pos, x / y
        sq(log[:, 20], log[:, 21]), sq(log[:, 22], log[:, 21]),                    # Fz, running mean Fz vs goal
        sq(log[:, 23], log[:, 24]),                                                # dFz/dt vs goal
        sq(speed, log[:, 9]), sq(log[:, 10], log[:, 9]),                           # |v|, running mean speed vs goal
        log[:, 19],                                                                # quaternion distance
        log[:, 41], log[:, 42], log[:, 44], log[:, 45], log[:, 43],                # pos, ori, force, derivative_force, vel reward terms
    ], dim=1)


class DeviceErrorMetrics:
    """Per-environment accumulators of the reference's error metrics; everything stays on the simulator's device.

        em = DeviceErrorMetrics(env)                  # turns the step log on
        obs, rew, done = env.step_tensor(actions); em.update(done)
        em.last[i]            # the 13 metrics of environment i's most recently finished episode (em.episodes[i] of them so far)
        em.mean()             # over all finished episodes of all environments
        em.save(i, "tracking")    # error_data/tracking/<metric>.csv, the files error.py writes"""

    def __init__(self, vec_env):
        self.env = vec_env
        if vec_env.step_log is None:
            vec_env.enable_step_log(True)
        n, dev = vec_env.num_envs, vec_env.device
        self.horizon = float(vec_env.horizon)
        self.acc = torch.zeros(n, len(METRICS), dtype=torch.float64, device=dev)
        self.last = torch.zeros_like(self.acc)
        self.total = torch.zeros(len(METRICS), dtype=torch.float64, device=dev)
        self.episodes = torch.zeros(n, dtype=torch.int64, device=dev)

    def update(self, done):
        """after every step; `done`: that step's done flags ([n] bool / uint8 tensor on the device)"""
        self.acc += step_terms(self.env.step_log.to(torch.float64))
        d = done.to(torch.bool)
        fin = self.acc / self.horizon
        self.last = torch.where(d[:, None], fin, self.last)
        self.total += (fin * d[:, None]).sum(0)
        self.episodes += d
        self.acc *= (~d)[:, None]

    def mean(self):
        """{metric: mean over every finished episode of every environment}"""
        m = (self.total / max(int(self.episodes.sum()), 1)).cpu().numpy()
        return dict(zip(METRICS, m.tolist()))

    def save(self, env_index, model_name, root="."):
        """error.py:5-16 save_data: error_data/<model_name>/<metric>.csv, one value, no header"""
        fldr = os.path.join(root, "error_data", model_name)
        os.makedirs(fldr, exist_ok=True)
        vals = self.last[int(env_index)].cpu().numpy()
        for name, v in zip(METRICS, vals):
            with open(os.path.join(fldr, name + ".csv"), "w") as f:
                f.write(repr(float(v)) + "\n")
        return dict(zip(METRICS, vals.tolist()))


def metrics_from_csv(root, idx):
    """numpy restatement of error.py:148-191 over the files episode_log.EpisodeLogger writes (`<stem>_<idx>.csv`): the checker of the
    accumulators above, and a drop-in for calculate_error_metrics where pandas is not wanted"""
    load = lambda folder, stem: np.loadtxt(os.path.join(root, folder, f"{stem}_{idx}.csv"), delimiter=",", ndmin=2)
    sim = lambda stem: load("simulation_data", stem)
    rew = lambda stem: load("reward_data", stem)
    mse = lambda a, b: float(np.square(np.subtract(a, b)).mean())
    speed = np.linalg.norm(sim("ee_vel"), axis=1, keepdims=True)
    return {
        "x_pos_mse": mse(sim("ee_pos")[:, 0], sim("ee_goal_pos")[:, 0]), "y_pos_mse": mse(sim("ee_pos")[:, 1], sim("ee_goal_pos")[:, 1]),
        "force_mse": mse(sim("ee_z_contact_force"), sim("ee_z_goal_contact_force")),
        "mean_force_mse": mse(sim("ee_z_running_mean_contact_force"), sim("ee_z_goal_contact_force")),
        "der_force_mse": mse(sim("ee_z_derivative_contact_force"), sim("ee_z_goal_derivative_contact_force")),
        "velocity_mse": mse(speed, sim("ee_goal_vel")), "mean_velocity_mse": mse(sim("ee_running_mean_vel"), sim("ee_goal_vel")),
        "quat_diff_mean": float(sim("ee_diff_quat").mean()),
        "pos_reward_mean": float(rew("pos").mean()), "ori_reward_mean": float(rew("ori").mean()), "force_reward_mean": float(rew("force").mean()),
        "der_force_reward_mean": float(rew("derivative_force").mean()), "vel_reward_mean": float(rew("vel").mean()),
    }
