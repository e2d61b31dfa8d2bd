"""CPU oracle for the ssr_eval DSP/metric hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ssr_eval_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may.  The product path runs on the HIP library and fails
loudly when it is missing.

What this restates (all citations are into the reference tree
``haoheliu/ssr_eval`` v0.0.6):

* ``oracle.stft``      - ``librosa.stft`` as consumed at ``ssr_eval/metrics.py:27`` and
  ``ssr_eval/eval.py:29,37-38``; ``torchlibrosa.stft.STFT/ISTFT`` as consumed at
  ``ssr_eval/dsp.py:21-39,64,69,73,77,112``; ``librosa.istft`` (``eval.py:40``).
* ``oracle.ssim``      - ``skimage.metrics.structural_similarity(win_size=7)`` as called
  at ``ssr_eval/metrics.py:131`` (no ``data_range`` -> 2.0 for float images).
* ``oracle.metrics``   - ``AudioMetrics`` (``ssr_eval/metrics.py:15-132``) and the numeric
  helpers of ``ssr_eval/utils.py:43-92``.
* ``oracle.resample``  - ``librosa.resample(res_type="polyphase")`` (``eval.py:145-150``) ==
  ``scipy.signal.resample_poly``; SciPy is installed and is used directly as the pin.
* ``oracle.lowpass``   - ``ssr_eval/lowpass.py`` (``stft_hard_lowpass_v0`` :17-28,
  ``subsampling`` :134-144, ``align_length`` :31-51, ``lowpass`` :156-196).
* ``oracle.aggregate`` - ``ssr_eval/eval.py:200-216`` + ``ssr_eval/utils.py:24-28``.

PARITY STATUS.  The reference ships no tests, golden vectors or fixtures for
this path.  The reference's *own* arithmetic (lsd, sispec, to_log, energy
helpers, align_length, subsampling, cut-bin and key arithmetic, aggregation) is
pinned by importing ``/root/reference/ssr_eval`` in the build container and
committing its outputs under ``tests/golden/`` (``tests/golden/make_golden.py``).
The third-party primitives it delegates to are absent from the reference tree
and from this image: librosa (unpinned, effectively < 0.10), torchlibrosa
(>= 0.0.7), scikit-image (unpinned, a release that still infers data_range).
Their published algorithms are restated here and cross-checked against
independent implementations available in the image (``torch.stft``,
``torch.istft``, ``scipy.ndimage``, ``scipy.signal``), but **at those three
boundaries parity is unpinned**.
"""
