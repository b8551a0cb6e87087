"""MI355X-native audio-captioning forward/decode path (Cnn14Rnn-Trm), behind the plugin API of
wsntxxn/AudioCaption.  See DESIGN.md / INTEGRATION.md."""
from .cnn_encoder import Cnn14Encoder
from .config import cnn14rnn_trm_config, effb2_trm_config, init_model_from_config
from .effnet_encoder import EfficientNetB2
from .crnn_trm_encoder import Cnn14RnnEncoder, CrnnEncoder
from .rnn_encoder import RnnEncoder
from .transformer_decoder import TransformerDecoder
from .transformer_model import CaptionModel, TransformerModel

__all__ = ["Cnn14Encoder", "RnnEncoder", "CrnnEncoder", "Cnn14RnnEncoder", "TransformerDecoder",
           "CaptionModel", "TransformerModel", "init_model_from_config", "cnn14rnn_trm_config"]
