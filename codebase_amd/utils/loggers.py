"""Logger with the reference's interface (marlbase/utils/loggers.py:14-169): `log_metrics(list[dict])`,
`watch(model)`, `get_state()`, `info/warning/...`; FileSystemLogger appends one row per evaluation to
`results.csv` and writes `config.yaml`, with the same aggregation rule (`squash_info`: keys seen once
pass through, otherwise mean_/std_ of the per-entry SUMS) so utils/postprocessing keeps working.
Host-side I/O only - nothing here is on the GPU path."""
import csv
import logging
import math
import os
import time
from datetime import timedelta

import numpy as np
import yaml


def squash_info(infos):
    keys = set(k for i in infos for k in i) - {"TimeLimit.truncated", "terminal_observation"}
    out = {}
    for key in keys:
        values = [d[key] for d in infos if key in d]
        if len(values) == 1:
            out[key] = values[0]
            continue
        sums = [np.array(v).sum() for v in values]
        head, _, tail = key.rpartition("/")
        prefix = head + "/" if head else ""
        out[f"{prefix}mean_{tail}"] = np.mean(sums)
        out[f"{prefix}std_{tail}"] = np.std(sums)
    return out


class Logger:
    def __init__(self, project_name="marlhip", cfg=None):
        self._total_steps = cfg["algorithm"]["total_steps"] if cfg else 0
        self._start = time.time()
        self._prev_time = None
        self._prev = (0, 0)

    def log_metrics(self, metrics):
        d = squash_info(metrics)
        self.print_progress(d["updates"], d["environment_steps"], d["mean_episode_returns"], len(metrics) - 1)

    def print_progress(self, updates, steps, mean_returns, episodes):
        self.info(f"Updates {updates}, Environment timesteps {steps}")
        now = time.time()
        if self._prev_time:
            wall = now - self._prev_time
            self.info(f"UPS: {(updates - self._prev[0]) / wall:.2f}, FPS: {(steps - self._prev[1]) / wall:.2f} (wall time)")
            elapsed = timedelta(seconds=math.ceil(now - self._start))
            self.info(f"Elapsed Time: {elapsed}")
        if self._total_steps:
            self.info(f"Completed: {100 * steps / self._total_steps:.2f}%")
        self._prev, self._prev_time = (updates, steps), time.time()
        self.info(f"Last {episodes} episodes with mean returns: {mean_returns:.3f}")
        self.info("-------------------------------------------")

    def watch(self, model):
        self.debug(model)

    def debug(self, *a, **k):
        return logging.debug(*a, **k)

    def info(self, *a, **k):
        return logging.info(*a, **k)

    def warning(self, *a, **k):
        return logging.warning(*a, **k)

    def error(self, *a, **k):
        return logging.error(*a, **k)

    def critical(self, *a, **k):
        return logging.critical(*a, **k)

    def get_state(self):
        return None


class FileSystemLogger(Logger):
    def __init__(self, project_name="marlhip", cfg=None):
        super().__init__(project_name, cfg)
        self.results_path, self.config_path = "results.csv", "config.yaml"
        self._cols = self._rows = None
        if cfg is not None:
            with open(self.config_path, "w") as f:
                yaml.safe_dump(_plain(cfg), f)

    def log_metrics(self, metrics):
        """One row per call under ONE header.  The reference writes the header from its first row and assumes every later row
        has the same keys (loggers.py:144-158) - true there because its first evaluation always follows an update; the vectorised
        loop can evaluate before the first update (no `loss` yet), so a key that appears later widens the header and the file is
        rewritten with the earlier rows' new cells empty (pandas reads them as NaN)."""
        d = squash_info(metrics)
        if self._cols is None and os.path.exists(self.results_path) and os.path.getsize(self.results_path):
            with open(self.results_path, newline="") as f:  # resumed run directory
                rows = list(csv.reader(f))
            self._cols, self._rows = rows[0], [dict(zip(rows[0], r)) for r in rows[1:]]
        if self._cols is None:
            self._cols, self._rows = [], []
        fresh = [k for k in d if k not in self._cols]
        row = {k: d[k] for k in d}
        self._rows.append(row)
        if fresh:
            self._cols = ["environment_steps"] + sorted(k for k in set(self._cols) | set(fresh) if k != "environment_steps")
            with open(self.results_path, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(self._cols)
                for r in self._rows:
                    w.writerow([r.get(c, "") for c in self._cols])
        else:
            with open(self.results_path, "a", newline="") as f:
                csv.writer(f).writerow([row.get(c, "") for c in self._cols])
        self.print_progress(d["updates"], d["environment_steps"], d["mean_episode_returns"], len(metrics) - 1)

    def get_state(self):
        import pandas as pd

        return pd.read_csv(self.results_path, index_col=0)


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x
