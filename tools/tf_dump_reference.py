"""OFF-BOX script (never run on the GPU box, needs tensorflow-gpu==2.5.0, tensorflow_addons==0.14.0,
tensorflow_probability==0.13.0, dm-sonnet): loads a weight file saved by M1.save_weights (Keras tensor layouts),
runs the reference's own layers on the stored input and dumps the outputs, so that someone with TF can close the
"parity unpinned" gap of oracle/m1_oracle.py.  Usage:
    PYTHONPATH=/path/to/reference/tf2.5/scripts python tools/tf_dump_reference.py weights.npz input.npy out.npz
Only the SE block and the gate are compared layer-wise here; the deterministic m1() branch of the reference needs
the two-line fix of SURVEY.md App. C-1 before it can run end to end.
"""
import sys

import numpy as np


def main(wpath, xpath, opath):
    import tensorflow as tf
    from model.unets.network_blocks import GridAttentionBlock3D, SEResNetBottleNeck
    W = dict(np.load(wpath))
    x = np.load(xpath).astype(np.float32)
    cp = dict(padding="same")
    blk = SEResNetBottleNeck(filters=W["core.serse1.conv4.bias"].shape[0], kernel_size=(1, 3, 3), strides=(1, 2, 2),
                             conv_params=cp, reduction=8)
    y = blk(x)                                            # builds the variables
    names = ["conv1", "norm1", "conv2", "norm2", "conv3", "norm3", "conv4", "norm4", "conv6", "conv7"]
    for n in names:
        layer = getattr(blk, n)
        if n.startswith("conv"):
            layer.set_weights([W[f"core.serse1.{n}.kernel"], W[f"core.serse1.{n}.bias"]])
        else:
            layer.set_weights([W[f"core.serse1.{n}.gamma"], W[f"core.serse1.{n}.beta"]])
    np.savez(opath, serse1=blk(x).numpy())


if __name__ == "__main__":
    main(*sys.argv[1:4])
