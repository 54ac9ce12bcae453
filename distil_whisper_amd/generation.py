"""Host side of `WhisperForConditionalGeneration.generate` for the MI355X engine: generation config, decoder prompt
assembly (language / task / timestamps / prompt_ids), length rules and the output conventions of the reference.

Reference behaviour (the arithmetic lives in the third-party `transformers` package, installed 5.15.0;
`TF:` = transformers/models/whisper/generation_whisper.py):
  * call sites: run_distillation.py:1524-1528 (`generate_step`), run_eval.py:690-739 (`gen_kwargs`: max_length,
    return_timestamps, num_beams, top_k=0, language/task for multilingual checkpoints, assistant_model, prompt_ids and
    the long-form thresholds), run_eval.py:806-844 (`benchmark_gen`: encoder_outputs + min/max_new_tokens),
    run_pseudo_labelling.py:861-996;
  * decoder prompt: TF:1455-1608 `_retrieve_init_tokens` (<|startoftranscript|>, language id, task id,
    <|notimestamps|> unless timestamps are requested), TF:1853-1918 `_prepare_decoder_input_ids` (prompt_ids in front);
  * lengths: TF:1920-1947 `_set_max_new_tokens_and_length`;
  * logits processors and their order: TF:1774-1812 (begin-suppress, suppress, timestamp rules);
  * return value: without `return_dict_in_generate` the generated tokens only (decoder prompt and EOS stripped,
    right-padded with pad_token_id, TF:913-957 + `_pad_to_max_length`); with it an object whose `.sequences` holds
    prompt + generated tokens as GenerationMixin returns them.
Everything here is integer host logic; the token loop itself runs on the GPU in decoding.GreedyDecoder.
"""
import copy

import torch

# language code -> name (the public Whisper language list; `language=` accepts a code, a name or "<|code|>")
_LANGS = ("en:english,zh:chinese,de:german,es:spanish,ru:russian,ko:korean,fr:french,ja:japanese,pt:portuguese,"
          "tr:turkish,pl:polish,ca:catalan,nl:dutch,ar:arabic,sv:swedish,it:italian,id:indonesian,hi:hindi,fi:finnish,"
          "vi:vietnamese,he:hebrew,uk:ukrainian,el:greek,ms:malay,cs:czech,ro:romanian,da:danish,hu:hungarian,ta:tamil,"
          "no:norwegian,th:thai,ur:urdu,hr:croatian,bg:bulgarian,lt:lithuanian,la:latin,mi:maori,ml:malayalam,cy:welsh,"
          "sk:slovak,te:telugu,fa:persian,lv:latvian,bn:bengali,sr:serbian,az:azerbaijani,sl:slovenian,kn:kannada,"
          "et:estonian,mk:macedonian,br:breton,eu:basque,is:icelandic,hy:armenian,ne:nepali,mn:mongolian,bs:bosnian,"
          "kk:kazakh,sq:albanian,sw:swahili,gl:galician,mr:marathi,pa:punjabi,si:sinhala,km:khmer,sn:shona,yo:yoruba,"
          "so:somali,af:afrikaans,oc:occitan,ka:georgian,be:belarusian,tg:tajik,sd:sindhi,gu:gujarati,am:amharic,"
          "yi:yiddish,lo:lao,uz:uzbek,fo:faroese,ht:haitian creole,ps:pashto,tk:turkmen,nn:nynorsk,mt:maltese,"
          "sa:sanskrit,lb:luxembourgish,my:myanmar,bo:tibetan,tl:tagalog,mg:malagasy,as:assamese,tt:tatar,haw:hawaiian,"
          "ln:lingala,ha:hausa,ba:bashkir,jw:javanese,su:sundanese,yue:cantonese")
LANGUAGES = dict(item.split(":") for item in _LANGS.split(","))
TO_LANGUAGE_CODE = {name: code for code, name in LANGUAGES.items()}
TO_LANGUAGE_CODE.update({"burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb",
                         "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si",
                         "castilian": "es", "mandarin": "zh"})
TASK_IDS = ("translate", "transcribe")

# what `generate` understands; anything else raises (the reference's GenerationMixin validates its kwargs as well)
_CONFIG_KEYS = ("max_length", "max_new_tokens", "min_new_tokens", "num_beams", "do_sample", "top_k", "top_p",
                "eos_token_id", "pad_token_id", "bos_token_id", "decoder_start_token_id", "suppress_tokens",
                "begin_suppress_tokens", "no_timestamps_token_id", "max_initial_timestamp_index", "prev_sot_token_id",
                "lang_to_id", "task_to_id", "is_multilingual", "return_timestamps", "language", "task",
                "forced_decoder_ids", "num_return_sequences", "use_cache", "output_scores", "return_dict_in_generate",
                "num_assistant_tokens", "prompt_condition_type", "length_penalty", "repetition_penalty",
                "no_repeat_ngram_size", "temperature", "early_stopping", "num_beam_groups")


class GenerationConfig:
    """Attribute bag with the fields of `model.generation_config` the reference reads and writes (run_distillation.py:
    1006-1049, run_eval.py:697; TF:640-700).  Unknown optional fields are simply absent (`hasattr` is False), which is
    what the reference's own checks rely on (e.g. `hasattr(generation_config, "is_multilingual")`)."""

    def __init__(self, **kw):
        self.max_length = 448
        self.max_new_tokens = None
        self.min_new_tokens = None
        self.num_beams = 1
        self.do_sample = False
        self.eos_token_id = None
        self.pad_token_id = None
        self.bos_token_id = None
        self.decoder_start_token_id = None
        self.suppress_tokens = None
        self.begin_suppress_tokens = None
        self.return_timestamps = False
        self.num_return_sequences = 1
        self.use_cache = True
        self.prompt_condition_type = "first-segment"
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_model_config(cls, config):
        g = cls()
        for k in ("eos_token_id", "pad_token_id", "bos_token_id", "decoder_start_token_id", "suppress_tokens",
                  "begin_suppress_tokens", "forced_decoder_ids"):
            v = getattr(config, k, None)
            if v is not None:
                setattr(g, k, copy.copy(v))
        ml = getattr(config, "max_length", None)
        if ml:
            g.max_length = ml
        return g

    @classmethod
    def from_any(cls, obj):
        """Copy of a GenerationConfig of this module, of `transformers`, or of a plain dict."""
        if obj is None:
            return cls()
        if isinstance(obj, cls):
            return copy.deepcopy(obj)
        d = obj if isinstance(obj, dict) else (obj.to_dict() if hasattr(obj, "to_dict") else vars(obj))
        g = cls()
        for k, v in d.items():
            if k.startswith("_") or k in ("transformers_version",):
                continue
            if v is None and not hasattr(g, k):
                continue
            setattr(g, k, copy.deepcopy(v))
        return g

    def to_dict(self):
        return {k: copy.deepcopy(v) for k, v in vars(self).items() if not k.startswith("_")}

    def __repr__(self):
        return f"GenerationConfig({self.to_dict()})"


class GenerateOutput:
    """`return_dict_in_generate=True` result: .sequences int64 [B, prompt + generated] (GenerateEncoderDecoderOutput)."""

    def __init__(self, sequences, scores=None):
        self.sequences = sequences
        self.scores = scores

    def __getitem__(self, k):
        return getattr(self, k)

    def keys(self):
        return [k for k in ("sequences", "scores") if getattr(self, k) is not None]


def language_to_id(language, gc):
    """TF:1465-1488."""
    language = language.lower()
    if language in gc.lang_to_id:
        token = language
    elif language in TO_LANGUAGE_CODE:
        token = f"<|{TO_LANGUAGE_CODE[language]}|>"
    elif language in TO_LANGUAGE_CODE.values():
        token = f"<|{language}|>"
    else:
        is_code = len(language) == 2
        raise ValueError(f"Unsupported language: {language}. Language should be one of:"
                         f" {list(TO_LANGUAGE_CODE.values()) if is_code else list(TO_LANGUAGE_CODE.keys())}.")
    if token not in gc.lang_to_id:
        raise ValueError(f"{token} is not supported by this specific model as it is not in the "
                         "`generation_config.lang_to_id`. (You should just add it to the generation config)")
    return gc.lang_to_id[token]


def set_language_and_task(gc, language, task, is_multilingual):
    """TF:1420-1453 (same checks, same messages)."""
    if is_multilingual is not None:
        if not hasattr(gc, "is_multilingual"):
            raise ValueError("The generation config is outdated and is thus not compatible with the `is_multilingual` "
                             "argument to `generate`. Please update the generation config.")
        gc.is_multilingual = is_multilingual
    if hasattr(gc, "is_multilingual") and not gc.is_multilingual:
        if task is not None or language is not None:
            raise ValueError("Cannot specify `task` or `language` for an English-only model. If the model is intended "
                             "to be multilingual, pass `is_multilingual=True` to generate, or update the generation "
                             "config.")
    if language is not None:
        if not hasattr(gc, "lang_to_id"):
            raise ValueError("The generation config is outdated and is thus not compatible with the `language` "
                             "argument to `generate`. Please update the generation config.")
        gc.language = language
    if task is not None:
        if not hasattr(gc, "task_to_id"):
            raise ValueError("The generation config is outdated and is thus not compatible with the `task` argument "
                             "to `generate`. Please update the generation config.")
        gc.task = task


def retrieve_init_tokens(gc, batch_size, detect_language=None):
    """TF:1455-1608: the forced decoder prefix per batch row as a python list of lists.  `detect_language()` is called
    (and must return one language token id per row) when the config knows languages but none was given."""
    task = getattr(gc, "task", None)
    language = getattr(gc, "language", None)
    init = [gc.decoder_start_token_id]
    if task is None and language is None:
        forced = getattr(gc, "forced_decoder_ids", None)
        if forced is not None and len(forced) > 0 and forced[0][0] == 1:
            forced = [list(f) for f in forced]
            i = 1
            while len(forced) > 0 and forced[0][0] == i:
                init.append(forced[0][1])
                forced = forced[1:]
                i += 1
            if len(forced) > 0:
                raise ValueError("You are using token ids in `forced_decoder_ids` that do not seem to correctly follow "
                                 f"the prompt pattern of Whisper. Make sure that {forced} has an entry for all "
                                 f"indices >= 1 and < {forced[0][0]}.")
    undefined_lang = len(init) <= 1 or init[1] is None
    if isinstance(language, (list, tuple)):
        if any(l is None for l in language):
            raise TypeError("Expected `language` to be `None`, a single string (e.g. `'en'`), or a list of strings "
                            "with length equal to the batch size. Got a list containing `None`.")
        if len(language) != batch_size:
            raise ValueError("When passing a list of languages, the length of the list must match the batch size. "
                             f"Expected length of {batch_size}, but got {len(language)} languages.")
        languages = list(language)
    elif language is None:
        languages = [None] * batch_size
    else:
        languages = [language]
    rows = [list(init) for _ in languages]
    lang_ids = None
    if language is not None:
        lang_ids = [language_to_id(l, gc) for l in languages]
    elif hasattr(gc, "lang_to_id") and undefined_lang:
        if detect_language is None:
            raise ValueError("language detection needs the model")
        lang_ids = [int(x) for x in detect_language()]
    if lang_ids is not None:
        for i in range(len(rows)):
            if len(rows[i]) > 1:
                rows[i][1] = lang_ids[i]
            else:
                rows[i].append(lang_ids[i])
    for i in range(len(rows)):
        if task is not None:
            if task not in TASK_IDS:
                raise ValueError(f"The `{task}` task is not supported. The task should be one of `{list(TASK_IDS)}`")
            rows[i].append(gc.task_to_id[task])
            # (the reference calls replace_or_add here and drops its result: the append above is the whole effect)
        elif language is not None and hasattr(gc, "task_to_id"):
            if not any(t in rows[i] for t in gc.task_to_id.values()):
                rows[i].append(gc.task_to_id["transcribe"])
        if not gc.return_timestamps and hasattr(gc, "no_timestamps_token_id") and \
                rows[i][-1] != gc.no_timestamps_token_id:
            rows[i].append(gc.no_timestamps_token_id)
        elif gc.return_timestamps and rows[i][-1] == getattr(gc, "no_timestamps_token_id", None):
            rows[i] = rows[i][:-1]
        rows[i] = [t for t in rows[i] if t is not None]
    if len(rows) == 1 and batch_size > 1:
        rows = [list(rows[0]) for _ in range(batch_size)]
    return rows


def resolve_lengths(gc, prompt_len, max_target_positions, explicit_max_length):
    """-> (max_new_tokens, min_new_tokens) for a decoder prompt of prompt_len tokens.  TF:1920-1947 plus
    GenerationMixin._prepare_generated_length: `max_new_tokens` wins over `max_length`; a `max_length` is extended by
    the (capped) prompt length because the reference counts it for the text after the forced prefix; the total never
    exceeds max_target_positions."""
    mnt = gc.max_new_tokens
    if (mnt or 0) + prompt_len > max_target_positions and mnt is not None:
        raise ValueError(
            f"The length of `decoder_input_ids`, including special start tokens, prompt tokens, and previous tokens, "
            f"is {prompt_len},  and `max_new_tokens` is {mnt}. Thus, the combined length of `decoder_input_ids` and "
            f"`max_new_tokens` is: {mnt + prompt_len}. This exceeds the `max_target_positions` of the Whisper model: "
            f"{max_target_positions}. You should either reduce the length of your prompt, or reduce the value of "
            f"`max_new_tokens`, so that their combined length is less than {max_target_positions}.")
    if mnt is None:
        # TF:1932-1940: `max_length` counts the text after the forced prefix -- it is extended by the (capped) prompt
        # length and bounded by max_target_positions; GenerationMixin then stops at that total length
        num_initial = min(max_target_positions // 2 - 1, prompt_len)
        max_length = min(gc.max_length + num_initial, max_target_positions)
        mnt = max_length - prompt_len
        if mnt <= 0:
            raise ValueError(f"Input length of decoder_input_ids is {prompt_len}, but `max_length` is set to "
                             f"{max_length}. This can lead to unexpected behavior. You should consider increasing "
                             "`max_length` or, better yet, setting `max_new_tokens`.")
    mn = gc.min_new_tokens or 0
    return int(mnt), int(min(mn, mnt))


def strip_and_pad(sequences, prompt_len, eos_token_id, pad_token_id):
    """The plain return value of the reference's `generate` (TF:1060-1093 + `_pad_to_max_length`): per row the tokens
    after the decoder prompt, with trailing pads and the final EOS removed, right-padded to the longest row."""
    rows = []
    for row in sequences.tolist():
        seq = row[prompt_len:]
        if len(seq) and pad_token_id is not None and seq[-1] == pad_token_id:
            n_pad = sum(1 for t in seq if t == pad_token_id)
            if pad_token_id == eos_token_id:
                n_pad -= 1
            if n_pad:
                seq = seq[:-n_pad]
        if len(seq) and eos_token_id is not None and seq[-1] == eos_token_id:
            seq = seq[:-1]
        rows.append(seq)
    width = max((len(r) for r in rows), default=0)
    out = torch.full((len(rows), width), pad_token_id if pad_token_id is not None else 0, dtype=torch.long,
                     device=sequences.device)
    for i, r in enumerate(rows):
        if r:
            out[i, : len(r)] = torch.as_tensor(r, dtype=torch.long, device=sequences.device)
    return out


def retrieve_segment(seq, timestamp_begin, seek_num_frames, time_offset=0.0, time_precision=0.02,
                     time_precision_features=0.01, input_stride=2):
    """`WhisperGenerationMixin._retrieve_segment` (TF:generation_whisper.py:1977-2075) on a list of token ids: split the
    tokens generated for one window at consecutive timestamp pairs ("end of segment" predictions) and say how many mel
    frames the window consumed.  -> (segments [{"start", "end", "tokens"}], segment_offset in frames)."""
    n = len(seq)
    ts = [t >= timestamp_begin for t in seq]
    single_ending = ts[-2:] == [False, True]
    pairs = [i + 1 for i in range(n - 1) if ts[i] and ts[i + 1]]
    if pairs:
        slices = list(pairs)
        if single_ending:
            slices.append(n)
        else:
            slices[-1] += 1            # keep the last timestamp in the last segment: it marks "no single ending"
        segments, last = [], 0
        for i, cur in enumerate(slices):
            sl = seq[last:cur]
            is_last = i == len(slices) - 1
            end_tok = sl[-1] if (not is_last or single_ending) else sl[-2]
            segments.append({"start": time_offset + (sl[0] - timestamp_begin) * time_precision,
                             "end": time_offset + (end_tok - timestamp_begin) * time_precision, "tokens": sl})
            last = cur
        if single_ending:
            offset = int(seek_num_frames)          # a single timestamp at the end: no speech after it
        else:
            offset = (seq[last - 2] - timestamp_begin) * input_stride   # resume at the last predicted end of segment
        return segments, int(offset)
    stamps = [t for t in seq if t >= timestamp_begin]
    last_pos = int(seek_num_frames * time_precision_features / time_precision)
    if stamps and stamps[-1] != timestamp_begin:
        last_pos = stamps[-1] - timestamp_begin
    return [{"start": time_offset, "end": time_offset + last_pos * time_precision, "tokens": list(seq)}], \
        int(seek_num_frames)
