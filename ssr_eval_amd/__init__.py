"""ssr_eval_amd - MI355X-native implementation of the ssr_eval DSP / metric hot path.

Same public names as the reference package (ssr_eval/__init__.py:1-2).  Importing the package never
touches the GPU; the first call that needs arithmetic loads ``libssrhip.so`` and raises if it is missing
or if no HIP device is visible - there is no CPU path.
"""
from .eval import SSR_Eval_Helper, BasicTestee  # noqa: F401
from .metrics import AudioMetrics  # noqa: F401
from .dsp import FDomainHelper  # noqa: F401
from .lowpass import lowpass, bandpass  # noqa: F401

__version__ = "0.1.0"


def test():
    """The reference's quick-start (ssr_eval/test.py:21-38): identity testee, setting_fft cutoff 12 kHz,
    eval at 48 kHz, 10 files per speaker, on ./datasets/vctk_test (which must already exist)."""
    class MyTestee(BasicTestee):
        def infer(self, x):
            return x

    helper = SSR_Eval_Helper(MyTestee(), test_name="unprocessed", input_sr=44100, output_sr=44100, evaluation_sr=48000,
                             setting_fft={"cutoff_freq": [12000]}, save_processed_result=True)
    return helper.evaluate(limit_test_nums=10, limit_test_speaker=-1)
