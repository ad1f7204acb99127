"""MI355X-native drop-in for the stage-1 (ViT-VQGAN / RQ-VAE tokenizer) path of thuanz123/enhancing-transformers.

Import paths mirror the reference (``enhancing.modules.stage1.vitvqgan.ViTVQ`` etc.) so the reference's yaml
``target:`` strings resolve to these classes unchanged.  All arithmetic runs in hand-written gfx950 HIP
kernels reached through the C ABI in ``include/enh_hip.h`` (``enhancing._C``); there is no eager fallback.
"""
__version__ = "0.1.0"
