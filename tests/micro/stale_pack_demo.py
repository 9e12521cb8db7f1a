"""One-off demonstration for VERDICT r02 item 1a: with the version bumps of FusedAdam.step / ema.update switched off and
ema.copy_to / restore writing through .data (the round-2 code), tests/test_gpu_optim.py's
test_packed_weights_follow_fused_updates_and_ema_swaps must FAIL (stale packed weights).  Prints the outcome."""
import sys
import torch

sys.path.insert(0, ".")
torch.autograd.graph.increment_version = lambda *_a, **_k: None
from diffusionnag_amd import ema as ema_lib  # noqa: E402


def copy_to(self, parameters):
    with torch.no_grad():
        torch._foreach_copy_([p.data for p in parameters if p.requires_grad], self.shadow_params)


def restore(self, parameters):
    with torch.no_grad():
        torch._foreach_copy_([p.data for p in parameters], [c.data for c in self.collected_params])


ema_lib.ExponentialMovingAverage.copy_to = copy_to
ema_lib.ExponentialMovingAverage.restore = restore
from tests import test_gpu_optim as T  # noqa: E402

try:
    T.test_packed_weights_follow_fused_updates_and_ema_swaps()
except AssertionError as e:
    print("round-2 behaviour reproduced, the test fails as it should:", str(e)[:200])
else:
    print("UNEXPECTED: the test passes without the version bumps")
    sys.exit(1)
