"""Audio-callback entry of the real-time front-end: the per-block function the reference GUI calls from
its sounddevice callback, `custom_infer` (evaluations/real-time-gui.py:32-49), over this package's InferenceWrapper.

Same contract: the prompt is (re)computed and the stream caches are rebuilt lazily, only when the reference name or the block
size (in 2048-sample frames) changed since the previous call (:36-47, GUI settings: max_prompt_frames = 64, encode window 64,
decode window 64, max_seq_frames 768, buffer_frames 32); every call then converts one block with `process_one_chunk` (:48).
The reference keeps that state in module globals (:27-28); `RealtimeSession` holds it per instance, and the module-level
`custom_infer` keeps one default session so the reference's call sites work unchanged.

The GUI's quick presets (:629-662, file configs/presets.json: name -> {description, alpha, block_frame, n_frame_delay}) set the
three arguments this entry takes from the GUI -- alpha (noise mixing of the speaker embeddings), block_frame (frames per audio
block = decode_chunk_frames) and n_frame_delay -- so they are part of this boundary: `load_presets` / `apply_preset` /
`RealtimeSession.run_block`.

The GUI itself (customtkinter / sounddevice, :61-1461) is a caller of this boundary and is not part of the engine.
"""
import dataclasses
import json

import numpy as np


@dataclasses.dataclass
class GuiSettings:
    """The three `gui_config` fields a preset writes (real-time-gui.py:645-660) with the GUI's start-up values."""
    alpha: float = 0.7
    block_frame: int = 1
    n_frame_delay: int = 2


def load_presets(path="configs/presets.json"):
    """real-time-gui.py:634-640: the JSON object of the file, {} if it cannot be read."""
    try:
        with open(path, "r") as f:
            return json.load(f)
    except Exception:
        return {}


def apply_preset(settings: GuiSettings, name: str, presets: dict) -> GuiSettings:
    """real-time-gui.py:641-662: "Custom" or an unknown name changes nothing; otherwise each of the three keys present in the preset
    overwrites the setting (block_frame / n_frame_delay as ints)."""
    if name == "Custom" or name not in presets:
        return settings
    p = presets[name]
    out = dataclasses.replace(settings)
    if "alpha" in p:
        out.alpha = float(p["alpha"])
    if "block_frame" in p:
        out.block_frame = int(p["block_frame"])
    if "n_frame_delay" in p:
        out.n_frame_delay = int(p["n_frame_delay"])
    return out


class RealtimeSession:
    def __init__(self):
        self.reference_wav_name = ""
        self.decode_chunk_frames = 0
        self.prefills = 0                    # how many times the prompt was rebuilt (tests, diagnostics)

    def custom_infer(self, model_set, reference_wav, new_reference_wav_name, input_wav, n_frame_delay=2, alpha=0.7):
        """reference_wav: float array at 44.1 kHz; input_wav: [2048 * k] block (torch tensor or array) -> converted block of the
        same kind and length (zeros while the decoder delay fills, infer_arvc.py:519-525)."""
        n = int(input_wav.shape[-1])
        assert n % 2048 == 0 and n > 0, "block size must be a whole number of 2048-sample frames"
        frames = n // 2048
        if self.reference_wav_name != new_reference_wav_name or self.decode_chunk_frames != frames:
            ref = np.asarray(reference_wav.detach().cpu().numpy() if hasattr(reference_wav, "detach") else reference_wav, dtype=np.float32)
            model_set.prefill_prompt(ref.reshape(-1), max_prompt_frames=64, delay=n_frame_delay, alpha=alpha)
            model_set.setup_stream_caches(encode_window_frames=64, decode_window_frames=64, max_seq_frames=768, buffer_frames=32,
                                          decode_chunk_frames=frames)
            self.reference_wav_name = new_reference_wav_name
            self.decode_chunk_frames = frames
            self.prefills += 1
        is_torch = hasattr(input_wav, "detach")
        block = input_wav.reshape(1, -1) if not is_torch else input_wav.reshape(1, -1)
        pred = model_set.process_one_chunk(block)
        return pred.squeeze() if is_torch else np.asarray(pred).reshape(-1)


    def run_block(self, model_set, reference_wav, reference_wav_name, input_wav, settings: GuiSettings):
        """One audio-callback block under the GUI's current settings (real-time-gui.py:1313-1330 passes gui_config.n_frame_delay and
        gui_config.alpha; block_frame fixes the block length the callback delivers)."""
        assert int(input_wav.shape[-1]) == 2048 * settings.block_frame, "the audio callback delivers block_frame frames per block"
        return self.custom_infer(model_set, reference_wav, reference_wav_name, input_wav, n_frame_delay=settings.n_frame_delay,
                                 alpha=settings.alpha)


_default = RealtimeSession()


def custom_infer(model_set, reference_wav, new_reference_wav_name, input_wav, n_frame_delay=2, alpha=0.7):
    """real-time-gui.py:32-49 with the module-global state of the reference (one default session)."""
    return _default.custom_infer(model_set, reference_wav, new_reference_wav_name, input_wav, n_frame_delay=n_frame_delay, alpha=alpha)
