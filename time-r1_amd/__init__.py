"""time-r1_amd: MI355X-native GRPO rollout-and-update engine behind the Time-R1 trainer API."""
__version__ = "0.1.0"
