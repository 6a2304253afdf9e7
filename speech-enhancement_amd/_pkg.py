"""MI355X-native CMGAN / SCP-GAN hot path.  Public surface = the reference's Python call surface
for this path."""
from . import _lib  # noqa: F401
from .generator import TSCNet  # noqa: F401
from .discriminator import Discriminator, LearnableSigmoid, batch_pesq  # noqa: F401
from .frontend import compressed_stft, uncompressed_istft, normalize_batch  # noqa: F401
from .train import train_gan, validate_gan, batch_stft, gan_step, set_pesq_provider  # noqa: F401
from .optim import build_optimizer, set_weight_decay, LARS, Lamb  # noqa: F401
from .utils import adjust_learning_rate, kaiming_init, save_checkpoint  # noqa: F401
from .config import get_config  # noqa: F401
from .inference import load_model, predict  # noqa: F401
from .criterion import build_criterion  # noqa: F401
from .frontend import disassemble_spectrogram, power_compress, power_uncompress  # noqa: F401
from .diffuse import DiffuSE, inference_schedule  # noqa: F401
from .diffuse import predict as predict_diffuse  # noqa: F401
from .tsc_diffusion import TSCNetDiffusion, predict_tsc, tsc_diffusion_step, tsc_diffusion_validation_loss, add_noise  # noqa: F401
