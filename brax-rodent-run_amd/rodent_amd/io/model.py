"""`brax.io.model.save_params / load_params` [REF brax_rodent_run_ppo.py:138,204-205] on torch.save.
The reference pickles `(normalizer_params, policy_params)`; here a plain dict of tensors."""
import torch


def save_params(path: str, params) -> None:
    normalizer, policy = params[0], params[1]
    blob = {"policy": policy.state_dict() if hasattr(policy, "state_dict") else policy}
    if normalizer is not None:
        blob["normalizer"] = {k: getattr(normalizer, k).detach().cpu() for k in ("count", "mean", "summed_variance", "std")}
    torch.save(blob, path)


def load_params(path: str, map_location="cpu"):
    from ..training import running_statistics
    blob = torch.load(path, map_location=map_location, weights_only=True)
    norm = blob.get("normalizer")
    if norm is not None:
        norm = running_statistics.RunningStatisticsState(**norm)
    return norm, blob["policy"]
