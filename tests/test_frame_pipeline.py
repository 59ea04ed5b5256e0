"""rt_frame_ctx: frames kept in flight through the C ABI (render contexts, streams, band tiles, gathered / image / page-locked
host buffers and events all created once).  K frames in flight must be the frame rendered alone, byte for byte, and the
per-frame calls must leave the device's free memory where it was."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,w,h,exchange", [("meshMedBunny", 320, 200, True), ("meshMedBunny", 320, 200, False),
                                               ("meshTransparentBunny", 160, 96, True), ("lambertBPHSpheres", 250, 130, True)])
def test_frames_in_flight_are_the_frame_rendered_alone(name, w, h, exchange):
    import torch
    n = torch.cuda.device_count() if exchange else 1      # every visible GPU of this process (1 on a one-GPU box)
    scene = rt.Scene("scenes/%s.json" % name, w, h, 3)
    dss = [rt.DeviceScene(scene, device=r) for r in range(n)]
    comms = rt.Comm.init_all(list(range(n))) if exchange else None
    want, _, _, _ = dss[0].render_host(want_ids=False)
    slots = 3
    pipe = rt.FramePipeline(dss, comms, band_rows=16, slots=slots)
    # One pass of frames in flight first: RCCL sets its channels up with the first transfers on a communicator, and the HIP
    # runtime sizes a new stream's spill memory with its first launch and grows its own pools of kernel-argument space and
    # signals until they fit what is kept in flight.  The second pass is the steady state: there the per-frame calls must
    # not move the free memory of any device.
    frames = 12

    def one_pass():
        done = 0
        for k in range(frames + slots):
            s = k % slots
            if k >= slots:
                rgb, ms = pipe.wait(s, copy=False)
                assert np.array_equal(rgb, want) and ms > 0, "frame %d differs from the frame rendered alone" % (k - slots)
                done += 1
            if k < frames:
                pipe.enqueue(s, rt.RT_RENDER_STRIPS)
        assert done == frames

    one_pass()
    free0 = []
    for r in range(n):
        torch.cuda.synchronize(r)
        free0.append(torch.cuda.mem_get_info(r)[0])
    one_pass()
    for r in range(n):
        torch.cuda.synchronize(r)
        assert torch.cuda.mem_get_info(r)[0] == free0[r], "device %d: memory was allocated or freed by the per-frame calls" % r
    # a frame left on the device: no host copy, the same bytes in the slot's device image
    pipe.enqueue(1, rt.RT_FRAME_ON_DEVICE)
    none, ms = pipe.wait(1)
    assert none is None and ms > 0
    dev_img = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
    import ctypes
    import os
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))   # (the runtime torch and the library share)
    rc = hip.hipMemcpy(ctypes.c_void_p(dev_img.data_ptr()), ctypes.c_void_p(pipe.device_image(1)), ctypes.c_size_t(h * w * 3), 3)   # 3: device to device
    assert rc == 0 and np.array_equal(dev_img.cpu().numpy(), want)
    # misuse is refused, not raced
    pipe.enqueue(0)
    with pytest.raises(rt.RtError):
        pipe.enqueue(0)
    pipe.wait(0)
    with pytest.raises(rt.RtError):
        pipe.wait(0)
    with pytest.raises(rt.RtError):
        pipe.enqueue(slots)
    pipe.close()
    if comms:
        for c in comms:
            c.close()
