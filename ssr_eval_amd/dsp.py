"""FDomainHelper - drop-in for ssr_eval.dsp.FDomainHelper (ssr_eval/dsp.py:6-183) on MI355X.

The reference wraps torchlibrosa's conv1d-DFT ``STFT`` / ``ISTFT`` modules; here every transform is one
call into libssrhip.so (``ssr_stft`` COMPLEX, ``ssr_magphase``, ``ssr_istft``).  Tensors keep the
reference's shapes: waveforms [B, C, n] (or [B, n] for the single-channel helpers), spectrograms
[B, C, T, F].  Results are returned on the device of the input tensor.
"""
import numpy as np
import torch

from . import backend as B


class FDomainHelper(torch.nn.Module):
    def __init__(self, window_size=2048, hop_size=441, center=True, pad_mode="reflect", window="hann",
                 freeze_parameters=True, subband=None, root=None, *, precision="f64", device=None, engine="conv"):
        super().__init__()
        self.center, self.pad_mode, self.window = bool(center), pad_mode, window
        self.subband = subband
        div = 1 if subband is None else int(subband)          # dsp.py:40-59
        self.n_fft, self.hop = window_size // div, hop_size // div
        self.precision, self._device = precision, device
        # "conv": torchlibrosa's own arithmetic (dense float32 DFT products; n_fft must be a multiple of 32) - the default, as in
        # ssr_eval_amd.lowpass; "segments": float64 FFT (the exact transforms, faster)
        self.engine = engine if self.n_fft % 32 == 0 else "segments"
        # Anything but the configuration the reference instantiates (center=True, "reflect", "hann" - lowpass.py:18) exists in
        # torchlibrosa's own construction only: Conv1d weights from the named window, F.pad(mode=pad_mode) if center.  That is the
        # conv engine with other tables (ssr_plan_create_ex).
        self._ex = (not self.center) or pad_mode != "reflect" or window != "hann"
        if self._ex:
            if self.n_fft % 32 or self.n_fft > 4096:
                raise NotImplementedError("center / pad_mode / window beyond the defaults need n_fft = 32 m <= 4096 (the conv engine)")
            if pad_mode not in ("reflect", "constant"):
                raise NotImplementedError("pad_mode %r (torch's 'reflect' and 'constant' are implemented)" % (pad_mode,))
            self.engine = "conv"

    def _window_array(self):
        """librosa.filters.get_window(window, n_fft, fftbins=True): scipy's window of that name, periodic (torchlibrosa STFT.__init__)."""
        if self.window == "hann":
            return None
        import scipy.signal
        return np.asarray(scipy.signal.get_window(self.window, self.n_fft, fftbins=True), dtype=np.float64)

    def _plan(self):
        if self._ex:
            name = self.window if isinstance(self.window, (str, tuple, float, int)) else repr(self.window)
            return B.get_plan_ex(self.n_fft, self.hop, name, self._window_array(), self.center, self.pad_mode, self._device)
        return B.get_plan(self.n_fft, self.hop, self.precision, self._device, lowpass_engine=self.engine)

    # ---- [B, n] helpers -------------------------------------------------------------------------------
    def _stft(self, x):
        """x [B, n] -> (real, imag) each [B, 1, T, F] on x's device."""
        if x.dim() != 2:
            raise ValueError("expected [batch, samples], got %s" % (tuple(x.shape),))
        re, im = B.stft(self._plan(), [x[b].float() for b in range(x.shape[0])], kind="complex", torch_style_pad=True)
        return torch.stack(re)[:, None].to(x.device), torch.stack(im)[:, None].to(x.device)

    def _istft(self, real, imag, length):
        """(real, imag) [B, 1, T, F] -> [B, length]."""
        if length is None:
            # ISTFT._trim_edges(length=None): y[n_fft//2 : -n_fft//2] when centred, everything otherwise
            length = self.hop * (real.shape[2] - 1) + (0 if self.center else self.n_fft)
        nb = real.shape[0]
        y = B.istft(self._plan(), [real[b, 0] for b in range(nb)], [imag[b, 0] for b in range(nb)], [int(length)] * nb)
        return torch.stack(y).to(real.device)

    def complex_spectrogram(self, input, eps=0.0):
        real, imag = self._stft(input)
        return torch.cat([real, imag], dim=1)                  # [B, 2, T, F]

    def reverse_complex_spectrogram(self, input, eps=0.0, length=None):
        return self._istft(input[:, 0:1, ...], input[:, 1:2, ...], length)

    def spectrogram(self, input, eps=0.0):
        real, imag = self._stft(input.float())
        mag, _, _ = B.magphase(real.to(self._plan().device), imag.to(self._plan().device), eps)
        return mag.to(input.device)

    def spectrogram_phase(self, input, eps=0.0):
        real, imag = self._stft(input.float())
        dev = self._plan().device
        mag, cos, sin = B.magphase(real.to(dev), imag.to(dev), eps)
        return mag.to(input.device), cos.to(input.device), sin.to(input.device)

    # ---- [B, C, n] API --------------------------------------------------------------------------------
    def wav_to_spectrogram_phase(self, input, eps=1e-8):
        outs = [self.spectrogram_phase(input[:, c, :], eps=eps) for c in range(input.shape[1])]
        return tuple(torch.cat([o[i] for o in outs], dim=1) for i in range(3))

    def spectrogram_phase_to_wav(self, sps, coss, sins, length):
        chans = [self._istft(sps[:, c:c + 1] * coss[:, c:c + 1], sps[:, c:c + 1] * sins[:, c:c + 1], length)[:, None]
                 for c in range(sps.size(1))]
        return torch.cat(chans, dim=1)

    def wav_to_spectrogram(self, input, eps=1e-8):
        return torch.cat([self.spectrogram(input[:, c, :], eps=eps) for c in range(input.shape[1])], dim=1)

    def spectrogram_to_wav(self, input, spectrogram, length=None):
        wavs = []
        for c in range(input.shape[1]):
            # torchlibrosa.magphase divides by clamp(mag, 1e-10): ssr_magphase clamps the POWER re^2 + im^2, so the
            # equivalent floor is 1e-20 (dsp.py:147-152)
            _, cos, sin = self.spectrogram_phase(input[:, c, :], eps=1e-20)
            wavs.append(self._istft(spectrogram[:, c:c + 1] * cos, spectrogram[:, c:c + 1] * sin, length))
        return torch.stack(wavs, dim=1)

    def wav_to_complex_spectrogram(self, input, eps=0.0):
        return torch.cat([self.complex_spectrogram(input[:, c, :], eps=eps) for c in range(input.shape[1])], dim=1)

    def complex_spectrogram_to_wav(self, input, eps=0.0, length=None):
        n = input.size(1) // 2
        return torch.cat([self.reverse_complex_spectrogram(input[:, 2 * i:2 * i + 2], eps=eps, length=length)[:, None]
                          for i in range(n)], dim=1)
