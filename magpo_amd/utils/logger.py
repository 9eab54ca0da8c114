"""Console + marl-eval JSON logging with the reference's keys (mava/utils/logger.py:40-155, 373-433, 475-481).
Neptune / TensorBoard are external services and out of scope; enabling them raises."""
from __future__ import annotations

import json
import os
from datetime import datetime
from enum import Enum
from typing import Any, Dict

import numpy as np


class LogEvent(Enum):
    ACT = "actor"
    TRAIN = "trainer"
    EVAL = "evaluator"
    ABSOLUTE = "absolute"
    MISC = "misc"


def describe(x):
    x = np.asarray(x)
    if x.ndim == 0:
        return x.item()
    return {"mean": float(np.mean(x)), "std": float(np.std(x)), "min": float(np.min(x)), "max": float(np.max(x))}


def _flatten(d: Dict[str, Any], prefix: str = "") -> Dict[str, float]:
    out = {}
    for k, v in d.items():
        key = f"{prefix}/{k}" if prefix else k
        if isinstance(v, dict):
            out.update(_flatten(v, key))
        else:
            out[key] = v
    return out


class ConsoleLogger:
    def log_dict(self, data, step, eval_step, event):
        keys = ", ".join(f"{k.replace('_', ' ').title()}: {v:.3f}" if isinstance(v, float) else f"{k}: {v}" for k, v in data.items())
        print(f"{event.value.upper():<10} - {keys}", flush=True)

    def stop(self):
        pass


class JsonLogger:
    """marl-eval layout: {env}{task}{algo}{seed}{step_k: {step_count, metric: [values]}}."""
    _METRICS_TO_LOG = ("episode_return/mean", "win_rate", "steps_per_second")

    def __init__(self, base_exp_path, unique_token, system_name, path, task_name, env_name, seed):
        d = os.path.join(base_exp_path, "json", path) if path else os.path.join(base_exp_path, "json", system_name, unique_token)
        os.makedirs(d, exist_ok=True)
        self.file = os.path.join(d, "metrics.json")
        self.env, self.task, self.algo, self.seed = env_name, task_name, system_name, f"seed_{seed}"
        self.data: Dict[str, Any] = {}
        if os.path.exists(self.file):
            try:
                with open(self.file) as f:
                    self.data = json.load(f)
            except json.JSONDecodeError:   # cannot happen with the atomic writes below; a file from an older run may be truncated
                os.replace(self.file, self.file + ".corrupt")
        self.run = self.data.setdefault(self.env, {}).setdefault(self.task, {}).setdefault(self.algo, {}).setdefault(self.seed, {})

    def log_dict(self, data, step, eval_step, event):
        if event not in (LogEvent.EVAL, LogEvent.ABSOLUTE):
            return
        for key, value in data.items():
            if key not in self._METRICS_TO_LOG:
                continue
            k = "_".join(reversed(key.split("/"))) if "/" in key else key
            if event == LogEvent.ABSOLUTE:
                self.run.setdefault("absolute_metrics", {})[k] = [float(value)]
            else:
                st = self.run.setdefault(f"step_{eval_step}", {"step_count": int(step)})
                st[k] = [float(value)]
        tmp = self.file + ".tmp"   # write + rename: a kill during the write leaves the previous metrics.json intact
        with open(tmp, "w") as f:
            json.dump(self.data, f, indent=1)
        os.replace(tmp, self.file)

    def stop(self):
        pass


class MavaLogger:
    def __init__(self, config):
        self.cfg = config
        lg = config.logger.loggers
        token = datetime.now().strftime("%Y%m%d%H%M%S")
        self.loggers = []
        if lg.console.get("enabled", True):
            self.loggers.append(ConsoleLogger())
        if lg.json.get("enabled", False):
            self.loggers.append(JsonLogger(config.logger.base_exp_path, token, config.logger.system_name, lg.json.get("path"),
                                           lg.json.task_name, lg.json.env_name, lg.json.seed))
        for ext in ("neptune", "tensorboard"):
            if ext in lg and lg[ext].get("enabled", False):
                raise NotImplementedError(f"{ext} logging is an external service and is not part of this build")

    def log_config(self, config=None):
        pass

    def log(self, metrics, t: int, t_eval: int, event: LogEvent):
        metrics = dict(metrics)
        metrics.pop("is_terminal_step", None)
        if event == LogEvent.TRAIN:
            metrics = {k: float(np.mean(np.asarray(v))) for k, v in metrics.items()}
        else:
            metrics = {k: describe(v) for k, v in metrics.items()}
        flat = _flatten(metrics)
        for lg in self.loggers:
            lg.log_dict(flat, t, t_eval, event)

    def stop(self):
        for lg in self.loggers:
            lg.stop()
