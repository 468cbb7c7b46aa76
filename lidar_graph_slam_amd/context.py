"""Device context (mi355_ctx): one per GPU / per thread of control."""
from __future__ import annotations

import ctypes as C
import sys

from . import _lib as L

_DEFAULT = {}


class Context:
    def __init__(self, device_id: int = 0):
        self._lib = L.lib()
        self.handle = self._lib.mi355_create(int(device_id))
        if not self.handle:
            raise L.Mi355Error(L.EHIP, f"mi355_create({device_id}) failed: no HIP device (there is no CPU fallback)")
        self.device_id = device_id

    def check(self, rc: int):
        if rc != L.OK:
            raise L.Mi355Error(rc, self._lib.mi355_last_error(self.handle).decode())

    def set_stream(self, hip_stream: int | None):
        """Run on a caller-owned HIP stream, e.g. ``torch.cuda.current_stream().cuda_stream`` — 0 is torch's (legacy) default
        stream and is used as such, so device inputs produced by torch kernels on it are ordered with the library's work.
        ``None`` goes back to a private non-blocking stream of the library."""
        if hip_stream is None:
            self.check(self._lib.mi355_set_stream(self.handle, None, 1))
        else:
            self.check(self._lib.mi355_set_stream(self.handle, C.c_void_p(int(hip_stream)), 0))

    def synchronize(self):
        self.check(self._lib.mi355_synchronize(self.handle))

    def set_grid_cell(self, cell: float):
        self.check(self._lib.mi355_set_grid_cell(self.handle, C.c_float(cell)))

    def device_info(self):
        name = C.create_string_buffer(128)
        cu = C.c_int(0)
        self.check(self._lib.mi355_device_info(self.handle, name, 128, C.byref(cu)))
        return name.value.decode(), cu.value

    def close(self):
        if self.handle:
            if not getattr(self, "_borrowed", False):  # (a lane's context belongs to its mi355_multi: multi.MultiDevice.lane_context)
                self._lib.mi355_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass


def default_context(device_id: int = 0) -> Context:
    if device_id not in _DEFAULT:
        _DEFAULT[device_id] = Context(device_id)
    return _DEFAULT[device_id]
