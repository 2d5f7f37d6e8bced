"""``metrics.py`` of the reference (MetricLogger / MetricComputation, :13-128) without Lightning: all
metrics of one prediction come out of ONE fused kernel (`rdm_depth_metrics_f64`); under data parallelism
the per-pixel sums and the valid-pixel count are all-reduced before the division, so every rank reports the
global value (the reference logs per-rank values, `self.log` without sync_dist - SURVEY.md 2.1-C)."""
import torch
import torch.distributed as dist

from . import _lib

# name -> (index into the kernel's sums, post-processing)
_SLOTS = {"delta1": 1, "delta2": 2, "delta3": 3, "mse": 4, "mae": 5, "log10": 6, "absrel": 7, "sqrel": 8,
          "rmse": 9}          # NB 'rmse' is the reference's RelativeMeanSquareError: mean sqrt((p-t)^2/t) (metrics.py:107-110,128)


class MetricComputation:
    def __init__(self, metrics):
        for m in metrics:
            if m not in _SLOTS:
                raise KeyError(f"metric '{m}' is not built (available: {sorted(_SLOTS)})")
        self.names = list(metrics)
        self.reset()

    def reset(self):
        self.count = 0
        self.sum = [0.0 for _ in self.names]

    def compute(self, pred, target, sync=True):
        if not pred.is_cuda:
            raise _lib.RdmError("metrics run on the GPU only")
        p = pred.detach().double().contiguous()
        t = target.detach().double().contiguous()
        out = torch.empty(10, dtype=torch.float64, device=p.device)
        _lib.check(_lib.lib().rdm_depth_metrics_f64(_lib.ptr(p), _lib.ptr(t), p.numel(), _lib.ptr(out), _lib.stream()))
        if sync and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(out)
        out = out.cpu()
        assert out[0] > 0, "invalid target!"
        # delta1-3 are float32 in the reference: `(maxRatio < 1.25 ** k).float().mean()` (metrics.py:79-89) - an exact integer count
        # divided in float32; the other metrics stay in the dtype of the inputs (float64 from validation_step)
        n = float(out[0])
        values = [float(torch.tensor(float(out[_SLOTS[m]]), dtype=torch.float32) / n) if m.startswith("delta") else float(out[_SLOTS[m]]) / n
                  for m in self.names]
        self.count += 1
        for i, v in enumerate(values):
            self.sum[i] += v
        return values

    def avg(self, metric):
        if isinstance(metric, int):
            return self.sum[metric] / self.count
        return self.sum[self.names.index(metric)] / self.count


class MetricLogger:
    """log_train / log_val / log_test return the dicts the reference returns; `records` replaces self.log."""

    def __init__(self, metrics, module=None):
        self.context = module
        self.computer = MetricComputation(metrics)
        self.records = []

    def _log(self, prefix, pred, target, extra=None):
        values = self.computer.compute(pred, target)
        result = dict(extra or {})
        for name, value in zip(self.computer.names, values):
            result[name] = value
            self.records.append((f"{prefix}{name}", value))
        return result

    def log_train(self, pred, target, loss):
        return self._log("train_", pred, target, {"loss": loss})

    def log_val(self, pred, target):
        return self._log("val_", pred, target)

    def log_test(self, pred, target):
        return self._log("", pred, target)

    def reset(self):
        self.computer.reset()
