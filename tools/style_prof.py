import os, sys
sys.path.insert(0, os.getcwd())
import bench
from streamvoiceanon_amd import audio_io
from streamvoiceanon_amd.synth_audio import synth_utterance
w = bench._wrapper_with_prompt_path()
wav = synth_utterance(7407, 2048 * 107 + 100)
ref16 = audio_io.resample(wav, w.sr, w.RESAMPLE_FREQ)
w.style_encoder.use_graphs = False
for _ in range(3):
    w.calculate_style_vec(ref16)
