from fairseq.tasks import LegacyFairseqTask, register_task


@register_task("speecht5")          # (a second registration of the name: must be neutralised by whoever imports this module)
class SpeechT5Task(LegacyFairseqTask):
    def load_dataset(self, split, epoch=1, combine=False, **kwargs):
        self.datasets[split] = ("reference data plane", split, epoch, self.args.data, len(self.dicts["text"]))
