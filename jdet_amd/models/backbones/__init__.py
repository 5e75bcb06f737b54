from .resnet import ResNet, Resnet50, Resnet101  # noqa: F401
