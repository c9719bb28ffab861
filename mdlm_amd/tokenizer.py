"""Tokenizer handling for the offline hot path.  The reference uses the GPT-2
BPE tokenizer from the HF hub (dataloader.py:488-522), which is unavailable
offline; only its *contract* matters to the sampler: vocab_size 50257,
bos = eos = 50256 and no mask token (so mask_index = vocab_size,
diffusion.py:85-90)."""
import os


class SyntheticTokenizer:
  """GPT-2-shaped stand-in: ids decode to `<id>` strings."""

  def __init__(self, vocab_size=50257, bos_token_id=50256, eos_token_id=50256,
               mask_token_id=None):
    self.vocab_size = vocab_size
    self.bos_token_id = bos_token_id
    self.eos_token_id = eos_token_id
    self.pad_token_id = eos_token_id
    self.mask_token_id = mask_token_id
    self.mask_token = None if mask_token_id is None else '[MASK]'

  def decode(self, ids):
    return ' '.join(f'<{int(i)}>' for i in ids)

  def batch_decode(self, batch):
    return [self.decode(row) for row in batch]


def _finish_tokenizer(tokenizer):
  """The post-processing of dataloader.py:497-520: GPT-2 tokenizers wrap every
  sentence in [BOS] ... [EOS]; bos / eos fall back to cls / sep; a pad token is
  added when there is none."""
  import tokenizers
  import transformers
  gpt2 = tuple(c for c in (getattr(transformers, 'GPT2TokenizerFast', None),
                           getattr(transformers, 'GPT2Tokenizer', None)) if c is not None)
  if gpt2 and isinstance(tokenizer, gpt2):
    tokenizer._tokenizer.post_processor = tokenizers.processors.BertProcessing(
      (tokenizer.bos_token, tokenizer.bos_token_id),
      (tokenizer.eos_token, tokenizer.eos_token_id))
  if tokenizer.bos_token is None:
    if tokenizer.cls_token is None:
      raise AttributeError(f'Tokenizer must have a bos_token or cls_token: {tokenizer}')
    tokenizer.bos_token = tokenizer.cls_token
  if tokenizer.eos_token is None:
    if tokenizer.sep_token is None:
      raise AttributeError(f'Tokenizer must have a eos_token or sep_token: {tokenizer}')
    tokenizer.eos_token = tokenizer.sep_token
  if tokenizer.pad_token is None:
    tokenizer.add_special_tokens({'pad_token': '[PAD]'})
  return tokenizer


def get_tokenizer(config):
  """A local tokenizer directory is loaded with transformers; anything else
  (hub names cannot be fetched offline) yields the synthetic stand-in."""
  path = config.data.tokenizer_name_or_path
  if isinstance(path, str) and os.path.isdir(path):
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(path)
    return _finish_tokenizer(tokenizer)
  if isinstance(path, str) and 'bert' in path:
    # bert-base-uncased contract (dataloader.py:488-522): [CLS]=101 as BOS,
    # [SEP]=102 as EOS, [PAD]=0, an in-vocabulary [MASK]=103
    tok = SyntheticTokenizer(vocab_size=int(config.data.get('vocab_size', 30522)),
                             bos_token_id=101, eos_token_id=102, mask_token_id=103)
    tok.pad_token_id = 0
    return tok
  return SyntheticTokenizer(vocab_size=int(config.data.get('vocab_size', 50257)))
