from .basic_decoders import BasicDecoder
