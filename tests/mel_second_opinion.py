"""An evaluation of the NeMo-flavoured featurizer written INDEPENDENTLY of oracle/fa_oracle.c and of oracle.mel_f64: the header of
the reference (Sources/FluidAudio/Shared/AudioMelSpectrogram.swift:4-17) names NeMo's AudioToMelSpectrogramPreprocessor as its
spec, i.e. torch.stft(n_fft 512, hop 160, win_length 400, symmetric Hann, center=True, pad_mode="constant") -> |X|^2 -> librosa's
Slaney-normalised mel bank (librosa.filters.mel: htk=False, norm="slaney") -> log(x + 2^-24), after x[t] - 0.97 x[t-1] pre-emphasis.

Everything here is float64 and comes from the published librosa / torch formulas, not from the Swift file:
  * slaney_bank_f64: librosa.filters.mel's construction (mel_frequencies -> ramps -> min(lower, upper) -> enorm);
  * nemo_logmel_f64: pre-emphasis, torch.stft, power, bank, log.
The reference differs from torch in ONE documented place: its frame count is 1 + (L + 112) / 160, one more than
torch's 1 + L / 160 whenever L mod 160 >= 48 — the extra frame is a window truncated by the end of the zero-padded buffer.  Zeros
appended AFTER the pre-emphasis make torch.stft produce that frame too, so every length can be compared.
Test infrastructure only (tests/test_oracle_mel.py, tests/test_gpu_mel.py)."""
import numpy as np
import torch


def hz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def mel_to_hz_slaney(m):
    m = np.asarray(m, np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_bank_f64(sr=16000, n_fft=512, n_mels=128, fmin=0.0, fmax=None):
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    return w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def nemo_logmel_f64(audio, frames, window=None, bank=None, preemph=0.97, log_floor=2.0 ** -24, n_fft=512, hop=160, win=400):
    """[frames, n_mels] float64.  `window` / `bank`: override the float64 Hann / Slaney tables (the reference DEFINES its tables in
    fp32, :553-642; feeding those isolates framing + STFT + log from the table rounding)."""
    x = torch.from_numpy(np.ascontiguousarray(audio, np.float32)).to(torch.float64)
    y = x.clone()
    if preemph != 0.0 and x.numel() > 1:
        y[1:] = x[1:] - float(np.float32(preemph)) * x[:-1]
    need = (frames - 1) * hop                       # torch gives 1 + len // hop frames: extend with zeros until `frames` exist
    if y.numel() < need:
        y = torch.cat([y, torch.zeros(need - y.numel(), dtype=torch.float64)])
    w = torch.hann_window(win, periodic=False, dtype=torch.float64) if window is None else torch.from_numpy(np.asarray(window, np.float64))
    spec = torch.stft(y, n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=True, pad_mode="constant", return_complex=True)
    power = (spec.real ** 2 + spec.imag ** 2).numpy()[:, :frames]           # [257, frames]
    fb = slaney_bank_f64(n_fft=n_fft) if bank is None else np.asarray(bank, np.float64)
    return np.log(fb @ power + float(np.float32(log_floor))).T
