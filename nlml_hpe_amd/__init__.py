"""nlml_hpe_amd -- MI355X (gfx950) implementation of the NLML_HPE batched-inference hot path.

Host layer above the C ABI of ``include/nlml_hpe.h`` (``libnlml_hpe_hip.so``, built in-tree by
``__graft_entry__.build()``); every compute call runs a hand-written HIP kernel -- there is no CPU fallback.

    from nlml_hpe_amd import load_model
    model = load_model("models", device="cuda:0", encoder_state_dict=...)   # or a scripted .pth of the reference
    yaw, pitch, roll = model(x)                  # x f32[B,1404] on the GPU -> three [B,1] tensors (radians)
    pose, valid = model.from_landmarks(raw, return_valid=True)              # raw FaceMesh landmarks f32[B,468,3]

Modules: ``ops`` (tensor-level operators, also ``torch.ops.nlml_hpe.*``), ``model``, ``TD_Tester`` (Tucker
objective and device-side Powell under the reference's function names), ``video``, ``metrics``, ``pipeline``
(host-resident batches), ``distributed`` (sharding + all-gather), ``weights`` (artefact layout, packing), ``synth``
(seeded synthetic inputs/weights).
"""

__all__ = ["load_model", "HIPPoseModel"]


def __getattr__(name):   # lazy: importing the package must not import torch or load the library
    if name in ("load_model", "HIPPoseModel"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
