"""Input front-end of the HIP backend: decoded frames -> gray pair-kernel operands, on the device.

Mirrors what the reference does on the host to every decoded frame (FF = FunscriptFlow.pyw):

    AsyncVideoReader._decode_frame    cv2.cvtColor(BGR2RGB) FF:182, cv2.resize(frame, (256, 256)) FF:185-186
    fetch_frames_optimized, non-VR    cv2.cvtColor(f, COLOR_RGB2GRAY)                               FF:1082
    fetch_frames_optimized, VR        cv2.resize(f, (512, 512)); f[256:, :256]; RGB2GRAY            FF:1076-1079

One `ffl_upload_frames_raw` call does all of it (k_frontend): the frame is sent as decoded and only the
crop window of the resized image is ever computed.  No CPU fallback: without the HIP library this raises.
"""
from . import _capi


def geometry(width, height, vr_mode=False):
    """(resize size, crop origin) that yield a `width` x `height` operand.  The reference's sizes are
    width = height = 256 (FF:1057); VR keeps the bottom-left quadrant of a 2x larger resize (FF:1076-1079)."""
    if vr_mode:
        return (2 * width, 2 * height), (0, height)
    return (width, height), (0, 0)


def upload_decoded(ctx, first_slot, frames, vr_mode=False, rgb_order=False):
    """frames: (h, w, 3) uint8 arrays as cv2.VideoCapture.read returns them (BGR; pass rgb_order=True for
    frames that already went through FF:182) -> frame slots first_slot.. of `ctx`."""
    if not isinstance(ctx, _capi.Context):
        raise TypeError("upload_decoded needs a funscript_flow_amd._capi.Context")
    resize, crop = geometry(ctx.width, ctx.height, vr_mode)
    ctx.upload_frames_raw(first_slot, list(frames), resize, crop, rgb_order)


class DecodedUploader:
    """`upload` hook for pipeline.PairEngine: feeds it decoded frames instead of gray operands."""

    def __init__(self, ctx, vr_mode=False, rgb_order=False):
        self.ctx, self.vr_mode, self.rgb_order = ctx, bool(vr_mode), bool(rgb_order)

    def __call__(self, first_slot, frames):
        upload_decoded(self.ctx, first_slot, frames, self.vr_mode, self.rgb_order)
