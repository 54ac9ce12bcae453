"""MI355X-native hot path of huggingface/distil-whisper (Whisper distillation step).  See DESIGN.md.

Public surface:
  ops_hip.HipOps                      ctypes binding of libdwamd.so (C ABI in include/dwamd.h); no CPU fallback
  engine.{WhisperDims, ParamStore, WhisperEngine}   flat parameter store + hand-written forward/backward
  distill.DistillationTrainer        teacher fwd + student fwd/bwd + RCCL all-reduce + fused clip/AdamW
  modeling.{WhisperFeatureExtractor, WhisperForConditionalGeneration}   reference-shaped drop-in classes
  collator.DataCollatorSpeechSeq2SeqWithPadding, student_init.student_from_teacher
"""
__version__ = "0.1.0"
