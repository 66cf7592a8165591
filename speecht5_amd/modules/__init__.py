from .multihead_attention import MultiheadAttention  # noqa: F401
from .transformer_layer import TransformerSentenceEncoderLayer, TransformerDecoderLayer  # noqa: F401
from .encoder import TransformerEncoder, RelativePositionalEncoding  # noqa: F401
from .decoder import TransformerDecoder  # noqa: F401
from .speech_encoder_prenet import SpeechEncoderPrenet, ConvFeatureExtractionModel  # noqa: F401
from .speech_encoder_postnet import SpeechEncoderPostnet  # noqa: F401
from .speech_decoder_prenet import SpeechDecoderPrenet  # noqa: F401
from .speech_decoder_postnet import SpeechDecoderPostnet  # noqa: F401
from .text_encoder_prenet import TextEncoderPrenet  # noqa: F401
from .text_decoder_prenet import TextDecoderPrenet  # noqa: F401
from .text_decoder_postnet import TextDecoderPostnet  # noqa: F401
from .gumbel_vector_quantizer import GumbelVectorQuantizer  # noqa: F401
