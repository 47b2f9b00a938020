"""build_image_tower (reference multimodal_encoder/builder.py:15-36): CLIP towers only — the SigLIP and
S2 variants are not used by the qwen distillation shells."""
from .clip_encoder import CLIPVisionConfig, CLIPVisionTower


def build_image_tower(image_tower_cfg, **kwargs):
    image_tower = getattr(image_tower_cfg, "mm_image_tower", getattr(image_tower_cfg, "image_tower", None))
    if isinstance(image_tower, CLIPVisionConfig) or (isinstance(image_tower, str) and
                                                     ("openai" in image_tower or "clip" in image_tower.lower())):
        return CLIPVisionTower(image_tower, args=image_tower_cfg, **kwargs)
    raise ValueError(f"Unknown image tower: {image_tower}")
