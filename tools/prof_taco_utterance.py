#!/usr/bin/env python3
"""One B=1 Tacotron-2 inference of N forced frames, three times after a warm-up (run under rocprofv3 --kernel-trace --stats to see
the per-utterance kernels next to the decode chain).  usage: prof_taco_utterance.py [frames] [symbols]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.tacotron import Tacotron  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = Tacotron(dict(synth.TACOTRON_HPARAMS), 80, num_speakers=2)
m.load_state_dict(synth.tacotron_state())
m = m.cuda().eval()
n_sym = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ids = (torch.arange(n_sym) % 78 + 2)[None].cuda()
m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, n
for _ in range(4):
    m.inference(ids, None)
torch.cuda.synchronize()
