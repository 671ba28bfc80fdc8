"""MI355X-native Tacotron-2 + WaveGlow hot path.

Importing the package sets one HIP runtime knob, unless the caller already chose a value: kernel arguments are placed in
device memory (``HIP_FORCE_DEV_KERNARG=1``) instead of host memory, so the scalar loads that fetch a kernel's argument block
do not cross PCIe.  The decode chain of Tacotron-2 is five short dependent launches per frame and the WaveGlow forward ~180
launches with 200-300 byte argument blocks; measured on MI355X: B=1 decode 76.8 -> 70.6 us/step, WaveGlow forward -1.4 %
(profiles/r02_summary.md).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the process:
import this package before touching the GPU (bench.py and the tests do); afterwards the setting is simply ignored.
"""
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
