from .model import TransformerTTS  # noqa: F401
