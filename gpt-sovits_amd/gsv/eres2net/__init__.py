"""Speaker-verification front-end of the v2Pro / v2ProPlus models (reference GPT_SoVITS/eres2net/): Kaldi fbank + ERes2NetV2."""
