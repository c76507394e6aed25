"""llm_bci_amd — MI355X-native NDT1/CTC hot path behind the llm_bci plugin surface."""
__version__ = "0.1.0"
