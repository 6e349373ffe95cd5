"""Host side of the input front-end (no GPU): geometry of the reference's two uses, loud failure."""
import numpy as np
import pytest

from funscript_flow_amd import frontend


def test_geometry_matches_reference_sizes():
    assert frontend.geometry(256, 256, False) == ((256, 256), (0, 0))          # FF:1057, FF:185-186
    assert frontend.geometry(256, 256, True) == ((512, 512), (0, 256))         # FF:1076-1079: f[256:, :256]


def test_upload_decoded_refuses_anything_but_a_device_context():
    with pytest.raises(TypeError):
        frontend.upload_decoded(object(), 0, [np.zeros((4, 4, 3), np.uint8)])
