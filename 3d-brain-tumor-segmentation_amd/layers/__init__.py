"""layers package -- mirrors layers/__init__.py of the reference."""
from . import decoder, downsample, encoder, group_norm, resnet, upsample, vae  # noqa: F401

__all__ = ['encoder', 'decoder', 'vae', 'resnet', 'group_norm', 'downsample', 'upsample']
