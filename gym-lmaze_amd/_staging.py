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
        self._graphs = {}

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
            pin = self._pin_for(name, t)
            pin.copy_(t, non_blocking=True)
            out[name] = pin.numpy()
        torch.cuda.current_stream(self.device).synchronize()
        return out

    def _pin_for(self, name, t):
        pin = self._pin.get(name)
        if pin is None or pin.shape != t.shape or pin.dtype != t.dtype:
            pin = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
            self._pin[name] = pin
        return pin

    def step(self, key, value, body):
        """One reference-style step as ONE hipGraph launch: action upload, whatever `body(action_dev)` enqueues
        (the step kernel, the xE render), and the copies of the tensors it returns ({name: device tensor}) back to
        page-locked memory -- five or six operations whose launch gaps otherwise make up half of the 60 us.
        The first two calls per `key` run eagerly (they allocate the buffers), the third captures, later ones
        replay.  Returns {name: numpy view of the page-locked mirror}, valid until the next step."""
        state = self._graphs.get(key, 0)
        if not isinstance(state, tuple):
            act = self.action(value)                         # eager upload (allocates the staging words once)
            if state < 2:                                    # eager rounds
                outs = body(act)
                self._graphs[key] = state + 1
                return self.fetch(**outs)
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            graph, views = torch.cuda.CUDAGraph(), {}
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    self._act_dev.copy_(self._act_pin, non_blocking=True)
                    for name, t in body(self._act_dev).items():
                        pin = self._pin_for(name, t)
                        pin.copy_(t, non_blocking=True)
                        views[name] = pin.numpy()
            cur.wait_stream(side)
            self._graphs[key] = (graph, views)
        graph, views = self._graphs[key]
        self._act_np[0] = value
        graph.replay()
        torch.cuda.current_stream(self.device).synchronize()
        return views
