"""Pinned host staging for the single-env drop-in classes (N = 1, reference-typed returns).

A reference-style `step()` hands back host values: the (C, G*E, G*E) float32 observation, reward, done.
Going through pageable memory costs one blocking copy per value; here the action goes up and the
observation(s) + the scalar state block come down as asynchronous copies through page-locked buffers,
with ONE stream synchronisation per call.  Nothing here is on the batched path (device tensors in,
device tensors out, no sync)."""
import torch


class HostStaging:
    def __init__(self, device):
        self.device = device
        self._pin = {}
        self._act_pin = self._act_np = self._act_dev = None

    def action(self, value):
        """int -> device int32[1], uploaded asynchronously from a page-locked word."""
        if self._act_pin is None:
            self._act_pin = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self._act_np = self._act_pin.numpy()
            self._act_dev = torch.empty(1, dtype=torch.int32, device=self.device)
        self._act_np[0] = value
        self._act_dev.copy_(self._act_pin, non_blocking=True)
        return self._act_dev

    def fetch(self, **tensors):
        """Device tensors -> numpy VIEWS of the page-locked mirrors (valid until the next fetch of the same
        name: copy what must outlive it).  One synchronisation for all of them."""
        out = {}
        for name, t in tensors.items():
            pin = self._pin.get(name)
            if pin is None or pin.shape != t.shape or pin.dtype != t.dtype:
                pin = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
                self._pin[name] = pin
            pin.copy_(t, non_blocking=True)
            out[name] = pin.numpy()
        torch.cuda.current_stream(self.device).synchronize()
        return out
