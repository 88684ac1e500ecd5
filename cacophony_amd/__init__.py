"""cacophony_amd: MI355X-native (gfx950) inference hot path for the Cacophony audio-text model.

Host side of the drop-in boundary: the reference's Python model API
(`get_audio_embedding` / `get_text_embedding` / `get_contrastive_logits`, plus the
pre-processing free functions) over the C-ABI library `libcaco_hip.so` (include/caco_hip.h).
Importing the package never needs a GPU; creating a model does, and fails loudly without the
HIP library.
"""
from .config import (AudioMAEConfig, AudioTransformerConfig, CACOConfig, DatasetConfig, MelConfig,
                     RobertaConfig, default_audio_config, default_caco_config, default_text_config)

__all__ = [
    "AudioMAEConfig", "AudioTransformerConfig", "CACOConfig", "DatasetConfig", "MelConfig", "RobertaConfig",
    "default_audio_config", "default_caco_config", "default_text_config",
]
