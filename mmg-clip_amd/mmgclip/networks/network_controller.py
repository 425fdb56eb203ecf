from .encoder import (BertEncoder, ConvNextBaseEncoder, ConvNextTinyEncoder, ResNet50Encoder, ViTB16Encoder)  # noqa: F401


def getNetworkClass(network_name):
    """name -> class lookup with the reference's error (mmgclip/networks/network_controller.py:3-18).
    As in the reference, `ConvNextTiny` (the offline TorchScript loader) is deliberately NOT resolvable here."""
    network_class = globals().get(network_name, None)
    if network_class is None:
        raise ValueError(f"Invalid network_name: {network_name}")
    return network_class
