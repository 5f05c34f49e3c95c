"""Device chooser.  The reference (finenvs/device_utils.py:4-9) silently falls back
to "cpu" when CUDA is missing; this package has no CPU path, so it raises instead."""
import torch


def set_device(device_id: int) -> str:
    if device_id < 0:
        raise RuntimeError("finenvs_amd runs on an MI355X only: device_id must be >= 0 (no CPU path)")
    if not torch.cuda.is_available():
        raise RuntimeError(
            "finenvs_amd: PyTorch-ROCm sees no GPU; the HIP hot path has no CPU fallback"
        )
    return f"cuda:{device_id}"
