"""hvrnet_amd -- MI355X-native HVRNet video-detection forward path."""
__version__ = '0.1.0'
